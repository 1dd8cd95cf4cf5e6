"""CLI argument dataclasses of the MI355X backends (same conventions as the reference's ``arguments_classes/``: field
prefix = ``config_prefix`` of the BackendSpec, ``*_gen_*`` fields are folded into ``gen_kwargs`` -
backend_registry.py:122-142; defaults mirror whisper_stt_arguments.py:5-52, parakeet_tdt_arguments.py:5-39,
qwen3_tts_arguments.py:5-134 where the option has a counterpart)."""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional


@dataclass
class MI355XWhisperSTTHandlerArguments:
    stt_model_name: str = field(default="openai/whisper-large-v3", metadata={"help": "Checkpoint directory / hub id (large-v3, distil-large-v3, tiny.en); 'synthetic:<dims>' uses seeded random weights."})
    stt_device: str = field(default="cuda:0", metadata={"help": "GPU whose session pool serves this pipeline unit."})
    stt_torch_dtype: str = field(default="bfloat16", metadata={"help": "bfloat16 (serving) or float32 (exact-f32 parity policy)."})
    stt_max_batch: int = field(default=128, metadata={"help": "Sessions batched per engine pass on one GPU (engine limit 128)."})
    stt_batch_window_ms: float = field(default=4.0, metadata={"help": "How long the pool waits for more sessions before launching a batch."})
    stt_batch_cap: int = field(default=0, metadata={"help": "Largest final-STT batch the pool forms (0 = the engine's max_batch).  Near saturation a cap of 16-32 bounds "
                                                            "the batch time and lets a backlog drain (bench.py runs with 32)."})
    language: Optional[str] = field(default="en", metadata={"help": "Target language, or 'auto' for on-device language detection."})
    stt_gen_max_new_tokens: int = field(default=128)
    stt_gen_num_beams: int = field(default=1)
    stt_gen_return_timestamps: bool = field(default=False)
    stt_gen_task: str = field(default="transcribe")


@dataclass
class MI355XParakeetSTTHandlerArguments:
    parakeet_tdt_model_name: str = field(default="nvidia/parakeet-tdt-0.6b-v3")
    parakeet_tdt_device: str = field(default="cuda:0")
    parakeet_tdt_compute_type: str = field(default="bfloat16")
    parakeet_tdt_max_batch: int = field(default=128)
    parakeet_tdt_batch_window_ms: float = field(default=4.0)
    parakeet_tdt_batch_cap: int = field(default=0, metadata={"help": "Largest final-STT batch the pool forms (0 = the engine's max_batch)."})
    parakeet_tdt_stt_workers: int = field(default=0, metadata={"help": "STT workers of the device's session pool (thread + HIP stream + engine replica each; 0 = leave the pool as "
                                                                       "it is).  With live transcription a pass is a chain of short launches: 3 workers overlap them (measured best)."})
    parakeet_tdt_max_seconds: float = field(default=30.0)
    language: Optional[str] = field(default="en")


@dataclass
class MI355XQwen3TTSHandlerArguments:
    qwen3_tts_model_name: str = field(default="Qwen/Qwen3-TTS-12Hz-1.7B-CustomVoice")
    qwen3_tts_device: str = field(default="cuda:0")
    qwen3_tts_dtype: str = field(default="bfloat16")
    qwen3_tts_speaker: Optional[str] = field(default=None)
    qwen3_tts_language: Optional[str] = field(default="auto")
    qwen3_tts_instruct: Optional[str] = field(default=None)
    qwen3_tts_streaming_chunk_size: int = field(default=8, metadata={"help": "Codec frames per streamed audio chunk (8 = 640 ms)."})
    qwen3_tts_max_new_tokens: int = field(default=1536)
    qwen3_tts_blocksize: int = field(default=512)
    qwen3_tts_max_batch: int = field(default=128)
    qwen3_tts_batch_window_ms: float = field(default=4.0)
    qwen3_tts_lanes: int = field(default=3, metadata={"help": "TTS turn-batches in flight per GPU (engine replicas sharing the weights, one HIP stream each)."})
    qwen3_tts_codec_context_frames: int = field(default=25)
    # pacing of the continuously batched frame loop (session_pool.py): the high-load configuration bench.py measures is
    #   --qwen3_tts_lanes 1 --qwen3_tts_max_batch 512 --qwen3_tts_lead_s 0.5 --qwen3_tts_admit_wait_ms 250 --stt_batch_cap 32
    qwen3_tts_lead_s: float = field(default=0.0, metadata={"help": "Continuation rounds run only when a session's delivered audio leads its playback by less than this "
                                                                 "many seconds (0 = eager: synthesise as fast as the device allows)."})
    qwen3_tts_admit_wait_ms: float = field(default=0.0, metadata={"help": "With pacing: a new request joins the running frame loop at the next due round or after "
                                                                        "this wait, whichever comes first (0 = at the next chunk boundary)."})
    # on-device sampler (no counterpart in the reference's arguments: its backend samples internally); folded into gen_kwargs
    qwen3_tts_gen_do_sample: bool = field(default=False, metadata={"help": "Sample codec tokens on the device instead of greedy decoding."})
    qwen3_tts_gen_temperature: float = field(default=0.9)
    qwen3_tts_gen_top_k: int = field(default=50)
    qwen3_tts_gen_top_p: float = field(default=1.0)
    qwen3_tts_gen_repetition_penalty: float = field(default=1.05)
    qwen3_tts_gen_seed: int = field(default=0)


@dataclass
class EchoLLMHandlerArguments:
    """``--llm echo``: replies with the user's own transcript (benchmarks / BASELINE config C1)."""

    echo_llm_prefix: str = field(default="", metadata={"help": "Text put in front of the echoed transcript."})
