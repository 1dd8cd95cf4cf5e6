#!/usr/bin/env python
"""bench.py - the hot path of BASELINE.json on N MI355X GPUs of one node.

One "step" = one pass of the implemented hot path over one batch of synthetic sessions per GPU:
  config C3 (default): Whisper-large-v3 STT, 64 concurrent 10 s utterances batched on each GPU:
      PCM (resident in HBM) -> HIP log-mel -> 32-layer encoder -> cross-K/V -> 32 greedy decode tokens.
Sessions are the unit of data parallelism: every rank runs its own session pool on its own GPU with a full weight
replica, no collective on the data path -> "scaling": "weak".  The only cross-rank traffic is the
barrier + max-reduce of the step time the contract prescribes.

Prints ONE JSON line (rank 0) with the `roofline` (dominant kernel, HIP events inside this process) and
`cpu_baseline` (the CPU oracle timed on this node's host cores on a bounded sample) objects.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16, MI355X_MICROARCH.md:42
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md:35


def shard_sessions(per_gpu: int, rank: int) -> list[int]:
    """Global session ids served by this rank's pool: rank r owns [r*per_gpu, (r+1)*per_gpu) - weak scaling, no exchange."""
    return list(range(rank * per_gpu, (rank + 1) * per_gpu))


def reduce_step_time(dt: float, device) -> float:
    """The contract's cross-rank step: MAX over ranks of the timed region (the only collective in the job)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dt
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--sessions", type=int, default=64, help="concurrent sessions per GPU")
    ap.add_argument("--utt-seconds", type=float, default=10.0)
    ap.add_argument("--decode-tokens", type=int, default=32)
    ap.add_argument("--model", default="large-v3", choices=["large-v3", "distil-large-v3", "tiny.en", "test-tiny"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="time every kernel family with HIP events (stderr)")
    return ap.parse_args()


def dims_for(name):
    from speech_to_speech_amd.synth_weights import WhisperDims

    return {"large-v3": WhisperDims.large_v3, "distil-large-v3": WhisperDims.distil_large_v3,
            "tiny.en": WhisperDims.tiny_en, "test-tiny": WhisperDims.test_tiny}[name]()


def cpu_baseline(dims_full, utt_seconds, decode_tokens, layer_sample=4):
    """The CPU oracle (torch-CPU float32 port of the same algorithm) on ONE session of the same workload.

    Bounded sample: the model is instantiated with `layer_sample` of its (identical-cost) encoder and decoder
    layers; the measured layer time is scaled to the full depth, log-mel / stem / logits are measured in full.
    """
    import copy

    import numpy as np
    import torch

    from oracle import logmel as OL
    from oracle import whisper as OW
    from speech_to_speech_amd.synth import synth_audio

    cores = min(os.cpu_count() or 1, 32)     # torch-CPU stops scaling (and thrashes) far below 256 threads
    torch.set_num_threads(cores)
    dims = copy.copy(dims_full)
    dims.enc_layers = min(dims_full.enc_layers, layer_sample)
    dims.dec_layers = min(dims_full.dec_layers, layer_sample)
    g = torch.Generator().manual_seed(0)
    d, f = dims.d_model, dims.ffn
    sd = {}

    def lin(name, o, i, bias=True):
        sd[name + ".weight"] = torch.randn(o, i, generator=g) * i ** -0.5
        if bias:
            sd[name + ".bias"] = torch.randn(o, generator=g) * 0.02

    def ln(name):
        sd[name + ".weight"] = torch.ones(d)
        sd[name + ".bias"] = torch.zeros(d)

    e = "model.encoder."
    sd[e + "conv1.weight"] = torch.randn(d, dims.n_mels, 3, generator=g) * (3 * dims.n_mels) ** -0.5
    sd[e + "conv1.bias"] = torch.zeros(d)
    sd[e + "conv2.weight"] = torch.randn(d, d, 3, generator=g) * (3 * d) ** -0.5
    sd[e + "conv2.bias"] = torch.zeros(d)
    sd[e + "embed_positions.weight"] = torch.zeros(1500, d)
    for l in range(dims.enc_layers):
        p = f"{e}layers.{l}."
        ln(p + "self_attn_layer_norm"); ln(p + "final_layer_norm")
        for n in ("q_proj", "v_proj", "out_proj"):
            lin(p + "self_attn." + n, d, d)
        lin(p + "self_attn.k_proj", d, d, bias=False)
        lin(p + "fc1", f, d); lin(p + "fc2", d, f)
    ln(e + "layer_norm")
    dd = "model.decoder."
    sd[dd + "embed_tokens.weight"] = torch.randn(dims.vocab, d, generator=g) * 0.3
    sd[dd + "embed_positions.weight"] = torch.randn(dims.max_target, d, generator=g) * 0.3
    for l in range(dims.dec_layers):
        p = f"{dd}layers.{l}."
        for n in ("self_attn_layer_norm", "encoder_attn_layer_norm", "final_layer_norm"):
            ln(p + n)
        for a in ("self_attn.", "encoder_attn."):
            for n in ("q_proj", "v_proj", "out_proj"):
                lin(p + a + n, d, d)
            lin(p + a + "k_proj", d, d, bias=False)
        lin(p + "fc1", f, d); lin(p + "fc2", d, f)
    ln(dd + "layer_norm")
    audio = synth_audio(int(utt_seconds * 16000), 0)
    with torch.no_grad():
        t0 = time.perf_counter()
        feats = torch.from_numpy(OL.whisper_logmel(audio, dims.n_mels).astype(np.float32))[None]
        t1 = time.perf_counter()
        enc0 = OW.encode(sd, feats, dims.n_heads, 0)                      # stem + final norm only
        t2 = time.perf_counter()
        enc = OW.encode(sd, feats, dims.n_heads, dims.enc_layers)
        t3 = time.perf_counter()
        OW.greedy_decode(sd, enc, dims.n_heads, 0, [[1, 2, 3, 4]], eos=-1, max_new_tokens=decode_tokens)   # embed + logits only
        t4 = time.perf_counter()
        OW.greedy_decode(sd, enc, dims.n_heads, dims.dec_layers, [[1, 2, 3, 4]], eos=-1, max_new_tokens=decode_tokens)
        t5 = time.perf_counter()
    t_mel, t_stem, t_enc, t_head, t_dec = t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4
    enc_layers_t = max(t_enc - t_stem, 0.0) * dims_full.enc_layers / max(dims.enc_layers, 1)
    dec_layers_t = max(t_dec - t_head, 0.0) * dims_full.dec_layers / max(dims.dec_layers, 1)
    est = t_mel + t_stem + enc_layers_t + t_head + dec_layers_t
    return {"value": utt_seconds / est, "unit": "concurrent real-time sessions (audio-s per wall-s)", "cores": cores,
            "kind": "port",
            "sample": f"1 of the batch's sessions (one {utt_seconds:g} s utterance, log-mel + encoder + {decode_tokens} greedy "
                      f"tokens) on the torch-CPU float32 oracle with {dims.enc_layers}/{dims_full.enc_layers} encoder and "
                      f"{dims.dec_layers}/{dims_full.dec_layers} decoder layers measured ({t5 - t0:.1f} s of CPU work) and the "
                      f"per-layer time scaled to full depth -> {est:.1f} s per utterance"}


def main():
    args = parse()
    import torch

    from speech_to_speech_amd import native
    from speech_to_speech_amd.frontend import pack_pcm
    from speech_to_speech_amd.synth import synth_audio
    from speech_to_speech_amd.whisper import WhisperEngine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    dims = dims_for(args.model)
    B = args.sessions
    eng = WhisperEngine.from_random(dims, torch.bfloat16, max_batch=B, device=dev, seed=rank)
    n = int(args.utt_seconds * 16000)
    audios = [synth_audio(n, sid) for sid in shard_sessions(B, rank)]
    pcm, ns = pack_pcm(audios, dev)            # inputs resident in HBM before the timed region
    lib = native.load()
    prompt = [[1, 2, 3, 4]] * B
    suppress_eos = [dims.vocab - 1]

    def step():
        feats = torch.empty((B, 3002, dims.n_mels), dtype=torch.bfloat16, device=dev)
        native.check(lib.s2s_logmel_whisper_tm(eng.frontend._handle, native.stream_handle(), native.ptr(pcm), pcm.stride(0),
                                               native.ptr(ns), B, native.ptr(feats), native.S2S_BF16))
        eng.encode(feats)
        ids, _ = eng.decode(prompt, eos=dims.vocab - 1, max_new_tokens=args.decode_tokens, suppress=suppress_eos)
        return ids

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    cats = {"gemm_tile": native.PROF_GEMM_TILE, "gemm_skinny": native.PROF_GEMM_SKINNY, "attn_mfma": native.PROF_ATTN_MFMA,
            "attn_rows": native.PROF_ATTN_ROWS}
    native.prof_enable(*(cats.values() if args.breakdown else [native.PROF_GEMM_TILE]))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ids = step()
    barrier()
    dt = time.perf_counter() - t0
    prof = native.prof_collect(native.PROF_GEMM_TILE)
    if args.breakdown and rank == 0:
        for k, c in cats.items():
            if k != "gemm_tile":
                r = native.prof_collect(c)
                print(f"[breakdown] {k}: {r['ms'] / args.steps:.3f} ms/step over {r['launches'] // args.steps} launches, "
                      f"{r['flops'] / max(r['ms'], 1e-9) / 1e9:.1f} TFLOP/s, {r['bytes'] / max(r['ms'], 1e-9) / 1e6:.1f} GB/s", file=sys.stderr)
        print(f"[breakdown] gemm_tile: {prof['ms'] / args.steps:.3f} ms/step over {prof['launches'] // args.steps} launches", file=sys.stderr)
    native.prof_enable()
    assert all(len(x) == args.decode_tokens for x in ids)
    dt = reduce_step_time(dt, dev)
    if rank != 0:
        return
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.utt_seconds / (dt / args.steps)
    achieved = prof["flops"] / (prof["ms"] * 1e-3) / 1e12 if prof["ms"] > 0 else 0.0
    line = {
        "metric": "concurrent real-time sessions/GPU; p50 time-to-first-audio (ms), 10s utterance",
        "value": round(value, 2), "unit": "concurrent real-time sessions (audio-seconds of 10 s utterances transcribed per wall-second), whole job",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"C3 STT leg: Whisper-{args.model} (random-init weights at true dims), {B} concurrent "
                               f"{args.utt_seconds:g} s sessions batched per GPU, log-mel + encoder + {args.decode_tokens} greedy tokens",
                   "sessions_per_gpu": B, "utterance_seconds": args.utt_seconds, "decode_tokens": args.decode_tokens,
                   "stages": ["logmel", "whisper_encoder", "whisper_decoder"], "tts": "not in this bench line yet",
                   "parallelism": f"session-pool x{world} (no collectives)"},
        "p50_ttfa_ms": None,
        "roofline": {"bound": "mfma", "kernel": "gemm_tile_kernel<bf16> (encoder projections + conv stem + cross-K/V)",
                     "achieved": round(achieved, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
                     "launches": prof["launches"], "kernel_ms_per_step": round(prof["ms"] / args.steps, 3),
                     "algorithmic_tflop_per_step": round(prof["flops"] / args.steps / 1e12, 3)},
    }
    if not args.no_cpu_baseline and world == 1:
        try:
            line["cpu_baseline"] = cpu_baseline(dims, args.utt_seconds, args.decode_tokens)
        except Exception as ex:  # pragma: no cover
            line["cpu_baseline"] = {"error": repr(ex)}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
