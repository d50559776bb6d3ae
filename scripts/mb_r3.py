"""Round-3 A/B of the option-LSTM step kernels (headline shape: 20 000 rows, H = 512, 20 steps), one process per library
build (VD_LIB_PATH), knobs through vd_tune_set:
  * occupancy: the LDS request padded so that 3 / 2 / 1 workgroups fit a CU (real epilogue and K-loop-only diagnostic)
    -> the matrix-pipe duty of one, two and three waves per SIMD;
  * XCD partition of the tiles (VD_GLDS_XMAP 0 / 1 / 2);
Prints ms and executed TFLOP/s (fp32 MFMA peak 157.3)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from visdial_amd import ops


def timeit(fn, iters=4, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = "cuda"
    T, N, H, V = 20, 20000, 512, 11322
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    Wh = rnd(H, 4 * H) * 0.04
    table = rnd(V + 1, 4 * H) * 0.1
    tok = torch.randint(0, V + 1, (T, N), device=dev, dtype=torch.int32, generator=g)
    gates = torch.empty(T, N, 4 * H, device=dev)
    h = torch.empty(T, N, H, device=dev)
    c = torch.empty(T, N, H, device=dev)
    dcw = torch.empty(N, H, device=dev)
    dh_last = rnd(N, H)
    fl = 2.0 * N * H * 4 * H * (T - 1)
    fwd = lambda: ops.lstm_forward(table, Wh, gates, h, c, T, N, H, 0, 4 * H, tok_gather=tok)
    bwd = lambda: ops.lstm_backward(Wh, gates, c, dcw, T, N, H, dh_last=dh_last)
    print("library: %s" % os.environ.get("VD_LIB_PATH", "default"))
    variants = [("baseline (3 WG/CU)", {})]
    if os.environ.get("MB_FULL", "1") == "2":
        variants += [("K loop only (3 WG/CU)", dict(VD_LSTM_FWD_EPI_SEQ=2)),
                     ("K loop + epilogue-sized traffic spread over it", dict(VD_LSTM_FWD_EPI_SEQ=3)),
                     ("K loop + fire-and-forget traffic (DMA gather + store)", dict(VD_LSTM_FWD_EPI_SEQ=4)),
                     ("256 x 128 tiles, 8 waves, 64 KB LDS (3 A + 2 B buffers), 2 WG/CU", dict(VD_LSTM_FWD_BM256=1)),
                     ("256 x 128 tiles, 48 KB LDS (2 + 2 buffers)", dict(VD_LSTM_FWD_BM256=2)),
                     ("K loop + fire-and-forget DMA gathers only", dict(VD_LSTM_FWD_EPI_SEQ=5)),
                     ("K loop + fire-and-forget stores only", dict(VD_LSTM_FWD_EPI_SEQ=6)),
                     ("baseline (repeat)", {})]
    if os.environ.get("MB_FULL", "1") == "1":
        variants += [
            ("K loop only (3 WG/CU)", dict(VD_LSTM_FWD_EPI_SEQ=2)),
            ("K loop + epilogue-sized traffic spread over it", dict(VD_LSTM_FWD_EPI_SEQ=3)),
            ("the same, 2 WG/CU", dict(VD_LSTM_FWD_EPI_SEQ=3, VD_GLDS_LDS_BYTES=70 * 1024)),
            ("2 WG/CU", dict(VD_GLDS_LDS_BYTES=70 * 1024)),
            ("2 WG/CU, K loop only", dict(VD_GLDS_LDS_BYTES=70 * 1024, VD_LSTM_FWD_EPI_SEQ=2)),
            ("1 WG/CU", dict(VD_GLDS_LDS_BYTES=100 * 1024)),
            ("1 WG/CU, K loop only", dict(VD_GLDS_LDS_BYTES=100 * 1024, VD_LSTM_FWD_EPI_SEQ=2)),
            ("XCD map 1 (column slices per XCD)", dict(VD_GLDS_XMAP=1)),
            ("XCD map 1, K loop only", dict(VD_GLDS_XMAP=1, VD_LSTM_FWD_EPI_SEQ=2)),
            ("XCD map 1, no K rotation", dict(VD_GLDS_XMAP=1, VD_GEMM_ROTATE=0)),
            ("XCD map 2 (2 x 4)", dict(VD_GLDS_XMAP=2)),
            ("XCD map 2, no K rotation", dict(VD_GLDS_XMAP=2, VD_GEMM_ROTATE=0)),
            ("no K rotation", dict(VD_GEMM_ROTATE=0)),
            ("baseline (repeat)", {}),
        ]
    if os.environ.get("MB_FULL", "1") == "5":
        variants += [("K loop only", dict(VD_LSTM_FWD_EPI_SEQ=2)),
                     ("K loop + VALU lump after it (160 exp + 160 rcp + 640 fma)", dict(VD_LSTM_FWD_EPI_SEQ=7)),
                     ("K loop + the same VALU spread over it", dict(VD_LSTM_FWD_EPI_SEQ=8)),
                     ("K loop only (repeat)", dict(VD_LSTM_FWD_EPI_SEQ=2)),
                     ("baseline (repeat)", {})]
    if os.environ.get("MB_FULL", "1") == "4":
        variants += [("3 + 3 LDS buffers (48 KB, 3 WG/CU)", dict(VD_LSTM_FWD_DEEP=1)),
                     ("baseline (repeat)", {}),
                     ("3 + 3 LDS buffers (repeat)", dict(VD_LSTM_FWD_DEEP=1))]
    if os.environ.get("MB_FULL", "1") == "3":
        variants += [("128 x 256 tiles, 4 waves x (32 x 256), 72 KB, 2 WG/CU", dict(VD_LSTM_FWD_NT8=1)),
                     ("128 x 256 tiles, 3 A + 2 B buffers (56 KB)", dict(VD_LSTM_FWD_NT8=2)),
                     ("128 x 256 tiles, K loop only", dict(VD_LSTM_FWD_NT8=1, VD_LSTM_FWD_EPI_SEQ=2)),
                     ("baseline (repeat)", {})]
    for name, knobs in variants:
        ops.tune_clear()
        for k, v in knobs.items():
            ops.tune_set(k, v)
        ms = timeit(fwd)
        ms2 = timeit(bwd, iters=3)
        print("  [%-36s] fwd %6.2f ms %6.1f TF (%.3f) | bwd %6.2f ms %6.1f TF (%.3f)" % (
            name, ms, fl / ms / 1e9, fl / ms / 1e9 / 157.3, ms2, fl / ms2 / 1e9, fl / ms2 / 1e9 / 157.3))
    ops.tune_clear()


if __name__ == "__main__":
    main()
