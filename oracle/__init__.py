"""CPU oracle for the MI355X STT+TTS hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE.  It restates, on the CPU (numpy float64 /
torch-CPU float32), the algorithms the reference's hot path executes inside its
third-party model packages, each function citing the file:line it follows.  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import it, and only as the checker - never as the thing measured or shipped.
The product (``speech-to-speech_amd/``) never imports it and has no CPU fallback.

Pinning status (see DESIGN.md "Oracle"):
  * The reference itself holds NO golden vectors or numerical assertions for this path
    (every model is faked in its tests), and none of its own modules can
    be imported offline.  The arithmetic lives in third-party packages:
      - transformers (>=4.57, container has 5.15.0): Whisper + Parakeet model code and
        feature extractors.  PRESENT in this container -> every oracle function for
        log-mel / Whisper / Parakeet is pinned against outputs generated from it by
        ``tests/golden/make_golden.py`` (committed fixtures under ``tests/golden/``).
      - nano-parakeet (>=0.2.0), faster-whisper/CTranslate2, faster-qwen3-tts /
        qwentts-cpp-python==0.3.1: ABSENT.  Qwen3-TTS is restated from the sibling
        architecture in transformers (qwen3_omni_moe talker / code predictor / Code2Wav)
        and pinned against that sibling; against the real Qwen3-TTS checkpoint the
        parity is UNPINNED.
"""
