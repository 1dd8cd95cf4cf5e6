"""One launch of each encoder projection shape on the tile GEMM (whatever S2S_TILE5 selects), for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes:
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o f -- python tools/tile5_pmc.py
    python tools/tile5_pmc.py --read out_fetch out_write     per-launch HBM bytes (raw and with the x2 of pmc_traffic.py) vs the algorithmic bytes
Launch order = SHAPES order (the counter CSV is sorted by dispatch id)."""
import csv
import glob
import os
import sys

SHAPES = (("enc.qkv", 96000, 3840, 1280, False), ("enc.o", 96000, 1280, 1280, True), ("enc.fc1", 96000, 5120, 1280, False), ("enc.fc2", 96000, 1280, 5120, True),
          ("cross.kv", 96000, 2560, 1280, False))


def read(d, counter):
    rows = []
    for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path, newline="")):
            if r["Counter_Name"] == counter and "gemm_tile" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), r["Kernel_Name"][:40]))
    return [v for _, v, _ in sorted(rows)]


if len(sys.argv) > 1 and sys.argv[1] == "--read":
    f, w = read(sys.argv[2], "FETCH_SIZE"), read(sys.argv[3], "WRITE_SIZE")
    for i, (name, M, N, K, f32out) in enumerate(SHAPES):
        alg_r = (M * K + N * K) * 2 + (M * N * 4 if f32out else 0)
        alg_w = M * N * (4 if f32out else 2)
        fr, wr = f[i] * 1024, w[i] * 1024
        print(f"{name:9s} FETCH_SIZE {fr / 1e6:8.1f} MB raw ({2 * fr / 1e6:8.1f} x2)  algorithmic reads {alg_r / 1e6:8.1f} MB | WRITE_SIZE {wr / 1e6:8.1f} MB  algorithmic writes {alg_w / 1e6:8.1f} MB"
              f" | (raw fetch + write) / algorithmic = {(fr + wr) / (alg_r + alg_w):.2f}, (2 x fetch + write) / algorithmic = {(2 * fr + wr) / (alg_r + alg_w):.2f}")
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from speech_to_speech_amd import native

lib = native.load()
dev = "cuda:0"
for name, M, N, K, f32out in SHAPES:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    Cout = torch.zeros(M, N, dtype=torch.float32 if f32out else torch.bfloat16, device=dev)
    bias = torch.zeros(N, device=dev)
    torch.cuda.synchronize()
    native.check(lib.s2s_op_gemm(native.stream_handle(), native.ptr(A), A.stride(0), native.ptr(W), native.ptr(bias), native.ptr(Cout) if f32out else None, native.ptr(Cout),
                                 Cout.stride(0), M, N, K, native.S2S_BF16, native.S2S_F32 if f32out else native.S2S_BF16, native.ACT_GELU if name == "enc.fc1" else 0, 0), "s2s_op_gemm")
    torch.cuda.synchronize()
