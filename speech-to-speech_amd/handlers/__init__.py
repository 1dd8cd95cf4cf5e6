"""MI355X drop-in handlers for the reference's STT / TTS stages (registered through boundary/registry.py)."""
