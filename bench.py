#!/usr/bin/env python
"""bench.py - the hot path of BASELINE.json on N MI355X GPUs of one node.

One "step" = one conversational turn for every session of one GPU's session pool, inputs resident in HBM:
    16 kHz PCM of the user utterances -> HIP log-mel -> STT encoder -> greedy decode -> (echo "LLM": the transcript
    is the reply) -> Qwen3-TTS talker + code predictor frames -> codec decoder, streamed every 8 frames ->
    24->16 kHz + int16 on device -> host chunks.
Workloads (BASELINE.json configs):
    c3 (default)  Whisper-large-v3 + Qwen3-TTS-1.7B dims, 64 concurrent 10 s sessions batched per GPU
    c2            Parakeet-TDT-0.6B + Qwen3-TTS, single session, 30 s audio
Sessions are the unit of data parallelism: every rank runs its own pool on its own GPU with a full weight replica and
no collective on the data path -> "scaling": "weak"; the only cross-rank traffic is the contract's
barrier + MAX-reduce of the step time.

Schedule (default): STT legs back to back on one thread / stream; finished turn-batches go to `--tts-lanes` TTS lanes
(engine instances sharing the weights; thread + HIP stream each) whose latency-bound decode kernels overlap with each other
and with the next STT leg on the device.  The TTS frame loop replays a captured hipGraph.  `--serial` runs the legs back to
back on one stream.

Prints ONE JSON line (rank 0) incl. `roofline` (the kernel family with the largest device time, measured with HIP events on
its own stream inside this process over one extra step of the same workload right after the timed region - events cannot be
recorded inside a graph replay) and `cpu_baseline` (CPU oracle on this node's host cores, bounded sample).
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
TURN_REPLY_SECONDS = 10.0


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
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c3", choices=["c3", "c2", "tiny"])
    ap.add_argument("--sessions", type=int, default=None, help="concurrent sessions per GPU (default: 64 for c3, 1 for c2)")
    ap.add_argument("--decode-tokens", type=int, default=32, help="Whisper greedy tokens per utterance (eos masked)")
    ap.add_argument("--stt-only", action="store_true")
    ap.add_argument("--serial", action="store_true", help="run STT and TTS of every step back to back on one stream (default: the TTS leg of step i "
                                                          "overlaps the STT leg of step i+1 on a second stream/thread, as the session pool does)")
    ap.add_argument("--tts-lanes", type=int, default=3, help="TTS turn-batches in flight at once: each lane is an engine instance (shared weights, own KV "
                    "cache / workspace) on its own thread + HIP stream; the decode kernels are latency-bound, so two lanes overlap on the device")
    ap.add_argument("--stt-lanes", type=int, default=1, help="STT turn-batches in flight at once (Whisper engine instances sharing the weights, "
                    "own encoder / KV workspace, thread and HIP stream each)")
    ap.add_argument("--stt-priority", type=int, default=0, help="1: run the STT legs on a high-priority HIP stream (measured: STT leg 1.05 -> 0.78 s, but in the saturated closed loop the batches then queue for a TTS lane: TTFA 1.6 -> 2.4 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prof-steps", type=int, default=1, help="extra steps after the timed region used for the HIP-event roofline sample")
    ap.add_argument("--no-graphs", action="store_true", help="launch every decode kernel on the stream instead of replaying captured hipGraphs")
    ap.add_argument("--breakdown", action="store_true", help="per-kernel-family and per-stage times on stderr")
    return ap.parse_args()


# --------------------------------------------------------------------------------------------- CPU baseline
def _rand_whisper_sd(dims, g):
    import torch

    d, f, sd = dims.d_model, dims.ffn, {}

    def lin(name, o, i, bias=True):
        sd[name + ".weight"] = torch.randn(o, i, generator=g) * i ** -0.5
        if bias:
            sd[name + ".bias"] = torch.randn(o, generator=g) * 0.02

    def ln(name):
        sd[name + ".weight"], sd[name + ".bias"] = torch.ones(d), torch.zeros(d)

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
    return sd


def cpu_baseline(dims_full, utt_seconds, decode_tokens, layer_sample=24):
    """The CPU oracle (torch-CPU float32 port of the same algorithm) on ONE session's STT leg of the same workload.

    Bounded sample: the model is instantiated with `layer_sample` of its (identical-cost) encoder and decoder layers; the
    measured per-layer time is scaled to the full depth, log-mel / stem / logits are measured in full.  The TTS leg of
    the turn is not included in the CPU number (its oracle at 1.7B dims needs minutes per utterance); the value is
    therefore an UPPER bound of what the CPU path sustains.
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
    sd = _rand_whisper_sd(dims, torch.Generator().manual_seed(0))
    audio = synth_audio(int(utt_seconds * 16000), 0)
    with torch.no_grad():
        t0 = time.perf_counter()
        feats = torch.from_numpy(OL.whisper_logmel(audio, dims.n_mels).astype(np.float32))[None]
        t1 = time.perf_counter()
        OW.encode(sd, feats, dims.n_heads, 0)                             # stem + final norm only
        t2 = time.perf_counter()
        enc = OW.encode(sd, feats, dims.n_heads, dims.enc_layers)
        t3 = time.perf_counter()
        OW.greedy_decode(sd, enc, dims.n_heads, 0, [[1, 2, 3, 4]], eos=-1, max_new_tokens=decode_tokens)   # embed + logits only
        t4 = time.perf_counter()
        OW.greedy_decode(sd, enc, dims.n_heads, dims.dec_layers, [[1, 2, 3, 4]], eos=-1, max_new_tokens=decode_tokens)
        t5 = time.perf_counter()
    t_mel, t_stem, t_enc, t_head, t_dec = t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4
    est = (t_mel + t_stem + max(t_enc - t_stem, 0.0) * dims_full.enc_layers / max(dims.enc_layers, 1) + t_head
           + max(t_dec - t_head, 0.0) * dims_full.dec_layers / max(dims.dec_layers, 1))
    return {"value": (utt_seconds + TURN_REPLY_SECONDS) / est, "unit": "concurrent real-time sessions (20 s turn / CPU seconds of its STT leg)",
            "cores": cores, "kind": "port",
            "sample": f"STT leg of 1 session ({utt_seconds:g} s utterance: log-mel + Whisper encoder + {decode_tokens} greedy tokens) on the "
                      f"torch-CPU float32 oracle, {dims.enc_layers}/{dims_full.enc_layers} encoder and {dims.dec_layers}/{dims_full.dec_layers} "
                      f"decoder layers measured ({t5 - t0:.1f} s of CPU work), per-layer time scaled to full depth -> {est:.1f} s per "
                      f"utterance; TTS leg excluded (upper bound for the CPU path)"}


# --------------------------------------------------------------------------------------------- main
def main():
    args = parse()
    import numpy as np
    import torch

    from speech_to_speech_amd import native
    from speech_to_speech_amd.frontend import pack_pcm
    from speech_to_speech_amd.synth import synth_audio
    from speech_to_speech_amd.synth_weights import ParakeetDims, Qwen3TTSDims, WhisperDims
    from speech_to_speech_amd.text import ByteTextFrontend, IdTextCodec
    from speech_to_speech_amd.tts import TTSEngine, TTSPrompt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal switch for boxes with fewer GPUs than ranks: all ranks share device 0 and rendezvous over gloo, so the
    # multi-process code path (init, barrier, MAX-reduce, rank-0 line) can be exercised on a 1-GPU box.  Never set by the driver.
    shared_dev = os.environ.get("S2S_BENCH_SHARE_DEVICE0") == "1"
    if shared_dev:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_dev:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    lib = native.load()
    native.graphs_enable(not args.no_graphs)

    if args.workload == "c3":
        B, utt_s, stt_name = args.sessions or 64, 10.0, "Whisper-large-v3"
        wdims, tdims = WhisperDims.large_v3(), Qwen3TTSDims.tts_1_7b()
    elif args.workload == "c2":
        B, utt_s, stt_name = args.sessions or 1, 30.0, "Parakeet-TDT-0.6B"
        wdims, tdims = None, Qwen3TTSDims.tts_1_7b()
    else:
        B, utt_s, stt_name = args.sessions or 4, 5.0, "Whisper-test-tiny"
        wdims, tdims = WhisperDims.test_tiny(), Qwen3TTSDims.test_tiny()
    n_frames = int(TURN_REPLY_SECONDS * 12.5 / 8 + 0.999) * 8          # 128 codec frames = 10.24 s reply, 16 streamed chunks
    sessions = shard_sessions(B, rank)
    audios = [synth_audio(int(utt_s * 16000), sid) for sid in sessions]
    pcm, ns = pack_pcm(audios, dev)                                     # inputs resident in HBM before the timed region

    if wdims is not None:
        from speech_to_speech_amd.whisper import WhisperEngine

        stt = WhisperEngine.from_random(wdims, torch.bfloat16, max_batch=B, device=dev, seed=rank)
        eos = wdims.vocab - 1
        prompt = [[1, 2, 3, 4]] * B

        n_stt = 1 if (args.serial or args.stt_only) else max(1, args.stt_lanes)
        stt_replicas = [stt] + [WhisperEngine(wdims, None, torch.bfloat16, B, dev, engine_tensors=stt._weights) for _ in range(n_stt - 1)]

        def run_stt(eng=stt):
            feats = torch.empty((B, 3002, wdims.n_mels), dtype=torch.bfloat16, device=dev)
            native.check(lib.s2s_logmel_whisper_tm(eng.frontend._handle, native.stream_handle(), native.ptr(pcm), pcm.stride(0), native.ptr(ns), B,
                                                   native.ptr(feats), native.S2S_BF16))
            eng.encode(feats)
            ids, _ = eng.decode(prompt, eos=eos, max_new_tokens=args.decode_tokens, suppress=[eos])
            return ids
    else:
        from speech_to_speech_amd.parakeet import ParakeetEngine, convert_parakeet_state_dict
        from speech_to_speech_amd.synth_weights import parakeet_state_dict

        pd = ParakeetDims.tdt_0_6b()
        stt = ParakeetEngine(pd, parakeet_state_dict(pd, seed=12, gain=1.0), dtype=torch.bfloat16, max_batch=B, max_seconds=utt_s, device=dev)
        T = int(utt_s * 100) + 1

        stt_replicas = [stt]

        def run_stt(eng=stt):
            feats, nf = stt.frontend.parakeet(pcm, ns, T, out_dtype=torch.bfloat16)
            stt.encode(feats, nf)
            toks, _, _ = stt.tdt_greedy()
            return [[t for t in tk if t != pd.blank] for tk in toks]

    tts = None
    tts_replicas = []
    if not args.stt_only:
        K = 1 if args.serial else max(1, args.tts_lanes)
        tts = TTSEngine.from_random(tdims, torch.bfloat16, max_batch=B, max_prompt=128, max_frames=n_frames, max_codec_frames=33, device=dev, seed=rank)
        tts_replicas = [tts] + [TTSEngine(tdims, None, torch.bfloat16, B, 128, n_frames, 33, None, dev, engine_tensors=tts._weights) for _ in range(K - 1)]
        g = torch.Generator().manual_seed(7)
        front = ByteTextFrontend((torch.randn(258, tdims.t_hidden, generator=g)).numpy())
    codec = IdTextCodec()
    stage = {"stt": 0.0, "tts_first_chunk": 0.0, "tts_total": 0.0}

    def run_tts_one(eng, ids, t0):
        ttfa, n_chunks = None, 0
        prompts = [TTSPrompt(front.embed(codec.decode(i)[:100], 128)) for i in ids]       # echo "LLM"
        for chunks in eng.stream(prompts, [n_frames] * len(ids), chunk_size=8, ctx_frames=25, min_frames=n_frames):
            if ttfa is None:
                ttfa = time.perf_counter() - t0
            n_chunks += 1
        assert n_chunks == n_frames // 8
        return ttfa

    def step():
        t0 = time.perf_counter()
        ids = run_stt()
        t1 = time.perf_counter()
        ttfa = run_tts_one(tts, ids, t0) if tts is not None else None
        t2 = time.perf_counter()
        stage["stt"] += t1 - t0
        stage["tts_total"] += t2 - t1
        stage["tts_first_chunk"] += (ttfa or 0.0) - (t1 - t0)
        return ttfa

    def run_steps(n):
        """n steps; unless --serial the TTS leg of a turn-batch runs on one of the TTS lanes (own thread + HIP stream) while the
        following turn-batches' STT legs (and the other lane's TTS leg) run."""
        if args.serial or tts is None:
            return [step() for _ in range(n)]
        import contextlib
        import queue
        import threading

        q: "queue.Queue" = queue.Queue(maxsize=1)
        out, err = [], []

        def tts_worker(eng):
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                    while True:
                        item = q.get()
                        if item is None:
                            return
                        ids, t0 = item
                        ta = time.perf_counter()
                        out.append(run_tts_one(eng, ids, t0))
                        stage["tts_total"] += time.perf_counter() - ta
            except BaseException as e:  # noqa: BLE001
                err.append(e)

        ths = [threading.Thread(target=tts_worker, args=(e,)) for e in tts_replicas]
        [t.start() for t in ths]
        left = [n]
        left_lock = threading.Lock()

        def stt_worker(eng, own_stream):
            try:
                torch.cuda.set_device(dev)
                # STT legs sit directly in every session's time-to-first-audio: their stream gets the high hardware-queue priority
                prio = -1 if args.stt_priority else 0
                ctx = torch.cuda.stream(torch.cuda.Stream(device=dev, priority=prio)) if (own_stream or args.stt_priority) else contextlib.nullcontext()
                with ctx:
                    while not err:
                        with left_lock:
                            if left[0] <= 0:
                                return
                            left[0] -= 1
                        t0 = time.perf_counter()
                        ids = run_stt(eng)
                        with left_lock:
                            stage["stt"] += time.perf_counter() - t0
                        while not err:
                            try:
                                q.put((ids, t0), timeout=1.0)
                                break
                            except queue.Full:
                                pass
            except BaseException as e:  # noqa: BLE001
                err.append(e)

        if len(stt_replicas) == 1:
            stt_worker(stt, False)
        else:
            sths = [threading.Thread(target=stt_worker, args=(e, True)) for e in stt_replicas]
            [t.start() for t in sths]
            [t.join() for t in sths]
        for _ in ths:
            while True:
                try:
                    q.put(None, timeout=1.0)
                    break
                except queue.Full:
                    if err and not any(t.is_alive() for t in ths):
                        break
        [t.join() for t in ths]
        if err:
            raise err[0]
        return out

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    for k in stage:
        stage[k] = 0.0
    cats = {"gemm_tile": native.PROF_GEMM_TILE, "gemm_skinny": native.PROF_GEMM_SKINNY, "attn_mfma": native.PROF_ATTN_MFMA, "attn_rows": native.PROF_ATTN_ROWS}
    barrier()
    t0 = time.perf_counter()
    ttfas = run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    timed_stage = dict(stage)
    # Roofline sample: args.prof_steps more steps of the SAME workload right after the timed region with every launch of
    # the four kernel families bracketed by HIP events on its own stream.  The decode loops replay captured hipGraphs in
    # the timed region; events cannot be recorded inside a replay, so the sample launches the same kernels on the stream.
    n_prof = max(1, args.prof_steps)
    native.prof_enable(*cats.values())
    run_steps(n_prof)
    torch.cuda.synchronize()
    prof = {k: native.prof_collect(c) for k, c in cats.items()}
    native.prof_enable()
    stage.update(timed_stage)
    dt = reduce_step_time(dt, torch.device("cpu") if shared_dev else dev)
    if rank != 0:
        return
    step_s = dt / args.steps
    turn_audio_s = utt_s + (0.0 if args.stt_only else n_frames / 12.5)
    value = world * B * turn_audio_s / step_s
    if args.breakdown:
        for k, r in prof.items():
            print(f"[breakdown] {k}: {r['ms'] / n_prof:.2f} ms/step, {r['launches'] // n_prof} launches, "
                  f"{r['flops'] / max(r['ms'], 1e-9) / 1e9:.1f} TFLOP/s, {r['bytes'] / max(r['ms'], 1e-9) / 1e6:.1f} GB/s", file=sys.stderr)
        for k, v in stage.items():
            print(f"[breakdown] stage {k}: {v / args.steps * 1e3:.1f} ms/step", file=sys.stderr)
    dom = max(prof, key=lambda k: prof[k]["ms"])
    r = prof[dom]
    if dom in ("gemm_tile", "attn_mfma"):
        ach = r["flops"] / max(r["ms"] * 1e-3, 1e-12) / 1e12
        roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4)}
    else:
        ach = r["bytes"] / max(r["ms"] * 1e-3, 1e-12) / 1e9
        roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
    names = {"gemm_tile": "gemm_tile_kernel<bf16> (encoder / codec projections and convolutions-as-GEMM, M > 64)",
             "gemm_skinny": "gemm_skinny_kernel<bf16> (weight-streaming decode projections, M <= 64 sessions)",
             "attn_mfma": "attn_mfma_kernel (encoder flash attention)", "attn_rows": "attn_rows_kernel (decode attention over the KV caches)"}
    roof.update({"kernel": names[dom], "traffic": None, "launches_per_step": r["launches"] // n_prof,
                 "kernel_ms_per_step": round(r["ms"] / n_prof, 2),
                 "algorithmic_tflop_per_step": round(r["flops"] / n_prof / 1e12, 3), "algorithmic_gb_per_step": round(r["bytes"] / n_prof / 1e9, 2),
                 "other_families_ms_per_step": {k: round(v["ms"] / n_prof, 2) for k, v in prof.items() if k != dom},
                 "event_pair_overhead_us_subtracted_per_launch": round(float(lib.s2s_prof_event_overhead_us()), 2),
                 "sampled_over": f"{n_prof} step(s) of the same workload run right after the timed region (stream launches; the timed region replays hipGraphs)"})
    # HBM traffic of the dominant family from the PMC counters (separate rocprofv3 --pmc passes over the decode-GEMM
    # microbenchmark, profiles/r01_pmc_skinny.json): measured bytes / algorithmic bytes, applied to this run's bytes per launch
    pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_skinny.json")
    if dom == "gemm_skinny" and os.path.exists(pmc_path):
        ratio = json.load(open(pmc_path))["hbm_traffic_over_algorithmic"]
        roof["traffic"] = round(ratio * r["bytes"] / max(r["launches"], 1))
        roof["traffic_note"] = (f"bytes per launch = {ratio:.3f} x algorithmic bytes per launch; ratio from FETCH_SIZE (x2, gfx950) + WRITE_SIZE over "
                                "the decode-GEMM microbenchmark (profiles/r01_pmc_skinny.json), not collected live")
    ttfa_ms = None if ttfas[0] is None else round(float(np.median(ttfas)) * 1e3, 1)
    line = {
        "metric": "concurrent real-time sessions/GPU; p50 time-to-first-audio (ms), 10s utterance",
        "value": round(value, 2),
        "unit": "concurrent real-time sessions, whole job (= sessions x seconds of turn audio [utterance in + reply out] / wall seconds per turn)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 2), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"{args.workload.upper()}: {stt_name} + {'Qwen3-TTS test-tiny' if args.workload == 'tiny' else 'Qwen3-TTS-1.7B'} dims (random-init weights at those dims), {B} concurrent "
                               f"{utt_s:g} s sessions batched per GPU, one turn each: STT ({args.decode_tokens if wdims else 'TDT greedy'} tokens) -> echo -> "
                               f"{n_frames} codec frames ({n_frames / 12.5:.2f} s reply) streamed in 8-frame chunks" + (" [STT only]" if args.stt_only else ""),
                   "sessions_per_gpu": B, "utterance_seconds": utt_s, "reply_seconds": 0.0 if args.stt_only else n_frames / 12.5,
                   "stages_ms_per_step": {k: round(v / args.steps * 1e3, 1) for k, v in stage.items()},
                   "tts_lanes": len(tts_replicas), "stt_lanes": len(stt_replicas), "decode_graphs": not args.no_graphs,
                   "schedule": "serial" if (args.serial or tts is None) else f"pipelined: STT legs back to back on one stream; TTS legs on {len(tts_replicas)} lane(s) (thread + HIP stream each) overlapping the following turn-batches",
                   "parallelism": f"session-pool x{world} (no collectives)"},
        "xrt": round(value, 2),          # audio seconds (utterance in + reply out) processed per wall second, whole job
        "p50_ttfa_ms": ttfa_ms,
        "ttfa_note": "all sessions of the pool end their utterance at t=0; TTFA = STT batch + TTS prefill + first 8-frame chunk + codec + resample + D2H",
        "roofline": roof,
    }
    if not args.no_cpu_baseline and world == 1 and wdims is not None:
        try:
            line["cpu_baseline"] = cpu_baseline(wdims, utt_s, args.decode_tokens)
        except Exception as ex:  # pragma: no cover
            line["cpu_baseline"] = {"error": repr(ex)}
    print(json.dumps(line))


if __name__ == "__main__":
    try:
        main()
    finally:
        try:
            import torch.distributed as _dist

            if _dist.is_available() and _dist.is_initialized():
                _dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
