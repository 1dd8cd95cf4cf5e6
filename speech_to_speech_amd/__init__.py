"""Import shim: the package sources live in ``speech-to-speech_amd/`` (the name the
project layout prescribes, which is not a valid Python identifier).  This stub makes
them importable as ``speech_to_speech_amd.<module>``."""

import os as _os

_SRC = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "speech-to-speech_amd")
__path__.append(_SRC)  # type: ignore[name-defined]
REPO_ROOT = _os.path.dirname(_SRC)
PACKAGE_DIR = _SRC

__version__ = "0.1.0"
