"""Single-GPU batched 1D / 2D C2C fp64 benchmarks of the kernels behind t0/t3, in the CSV schema of the reference's
templateFFT/batchTest (Test_1D.cpp:180-189, Test_2D.cpp; published tables templateFFT/csv/batch_result{1D,2D}.csv):

    X,Y,Z,Buffer,hip_time,GFlops,num_iter,bandwidth,max error

Input re = i+1, im = 0 scaled to keep fp64 exact (Test_1D.cpp:49-52), round-trip max error (Test_1D.cpp:169-176), HIP-event
timing over num_iter in-place launches, buffer ~1 GiB.  "bandwidth" follows the reference's definition
buffer * 2 * passes / time (GB/s), passes = number of kernel launches per transform.  Run on the GPU box:
    python tools/batch_bench.py profiles/r01
"""
import math
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from distributedfft_amd import _lib as L, api  # noqa: E402

BUF = 1 << 30  # bytes


def time_ms(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def bench_1d(x, iters=200):
    lib = L.load()
    batch = BUF // 16 // x
    n = x * batch
    a = (torch.arange(n, device="cuda", dtype=torch.float64) % 1024 + 1).to(torch.complex128).reshape(batch, x)
    ref = a.clone()
    s = torch.cuda.current_stream().cuda_stream
    ms = time_ms(lambda: lib.dfft_fft1d_rows(a.data_ptr(), a.data_ptr(), x, batch, 0, 1, s), iters)
    a.copy_(ref)
    lib.dfft_fft1d_rows(a.data_ptr(), a.data_ptr(), x, batch, 0, 1, s)
    lib.dfft_fft1d_rows(a.data_ptr(), a.data_ptr(), x, batch, 0, -1, s)
    torch.cuda.synchronize()
    err = float((a / x - ref).abs().max().item())
    ops = batch * 5.0 * x * math.log2(x)
    mb = n * 16 / 2 ** 20
    return [x, batch, 1, round(mb, 2), round(ms, 5), round(ops / (1e6 * ms), 2), iters, round(mb / 1.024 * 2 / ms, 1), err]


def bench_2d(x, y, iters=100):
    """FFT along X (contiguous) and Y (stride X) of Z matrices [Y][X]."""
    lib = L.load()
    z = max(1, BUF // 16 // (x * y))
    n = x * y * z
    a = (torch.arange(n, device="cuda", dtype=torch.float64) % 1024 + 1).to(torch.complex128).reshape(z, y, x)
    ref = a.clone()
    s = torch.cuda.current_stream().cuda_stream

    def fwd(d):
        if d > 0:
            lib.dfft_fft1d_rows(a.data_ptr(), a.data_ptr(), x, y * z, 0, d, s)
            lib.dfft_fft1d_cols(a.data_ptr(), a.data_ptr(), y, x, z, 0, d, s)
        else:
            lib.dfft_fft1d_cols(a.data_ptr(), a.data_ptr(), y, x, z, 0, d, s)
            lib.dfft_fft1d_rows(a.data_ptr(), a.data_ptr(), x, y * z, 0, d, s)

    ms = time_ms(lambda: fwd(1), iters)
    a.copy_(ref)
    fwd(1)
    fwd(-1)
    torch.cuda.synchronize()
    err = float((a / (x * y) - ref).abs().max().item())
    ops = z * 5.0 * x * y * math.log2(x * y)
    mb = n * 16 / 2 ** 20
    return [x, y, z, round(mb, 2), round(ms, 5), round(ops / (1e6 * ms), 2), iters, round(mb / 1.024 * 4 / ms, 1), err]


def main():
    out = Path(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out")
    out.mkdir(parents=True, exist_ok=True)
    hdr = "X,Y,Z,Buffer,hip_time,GFlops,num_iter,bandwidth,max error\n"
    with open(out / "batch_result1D.csv", "w") as f:
        f.write(hdr)
        for x in [25, 125, 64, 128, 256, 512, 1024, 2048, 96, 192, 384, 768, 100]:
            r = bench_1d(x)
            f.write(",".join(map(str, r)) + "\n")
            print("1D", r, flush=True)
    with open(out / "batch_result2D.csv", "w") as f:
        f.write(hdr)
        for x, y in [(2048, 2048), (2048, 1024), (2048, 512), (2048, 256), (2048, 128), (1024, 2048), (1024, 1024), (1024, 512),
                     (1024, 256), (1024, 128), (512, 2048), (512, 1024), (512, 512), (512, 256), (512, 128), (256, 2048),
                     (256, 1024), (256, 512), (256, 256), (256, 128), (128, 2048), (128, 1024), (128, 512), (128, 256),
                     (128, 128), (768, 768), (384, 192)]:
            r = bench_2d(x, y)
            f.write(",".join(map(str, r)) + "\n")
            print("2D", r, flush=True)


if __name__ == "__main__":
    main()
