"""Shared by potrf_trace_join.py / dist_trace_join.py: the flop model of one logical GEMM launch and the
merging of the kernel launches that belong to it.

A logical launch (one gpx_gemm_launch call after K-chunking) may run as several kernels: `part=main` plus a
separately launched last round (`part=tail`) or a split-K tail (`part=split`; its reduce kernel is a different
kernel name and is counted under "other kernels").  Every part is logged with the launch's full M, N, K, so
the flops are counted ONCE per logical launch and the parts' durations are added (they run back to back on one
stream).  Flops: 2 M N K, halved for a lower-only result (SYRK), halved for one triangular operand; a lower
result from two triangular operands of one matrix (lauum, W^T W) is n^3 / 3 = 2 n^3 / 6.  No rate computed
this way can exceed the 78.6 TFLOP/s fp64 MFMA peak."""

PEAK_TF = 78.6


def colblock_flops(M, N, K, nb):
    """Grouped trailing update (gpx_gemm_colblocks) on ONE rank (P = 1): column block c starts c * nb rows below the
    launch's first row and only the 128-blocks on or below its diagonal tile's diagonal are computed."""
    fl, c = 0.0, 0
    while c * nb < N:
        w = min(nb, N - c * nb)
        rows = M - c * nb
        if rows > 0:
            h = min(nb, rows)                               # the slab's diagonal tile: lower 128-blocks only
            nblk = (min(h, w) + 127) // 128
            diag = sum(min(128, h - 128 * i) * min(w, 128 * (i + 1)) for i in range(nblk))
            diag += max(0, h - 128 * nblk) * w
            fl += 2.0 * K * (diag + max(0, rows - h) * w)
        c += 1
    return fl


def launch_flops(M, N, K, lower, atri, btri, cb=0):
    if cb:
        return colblock_flops(M, N, K, cb)
    fl = 2.0 * M * N * K
    if lower and atri and btri:
        return fl / 6.0
    if lower:
        fl *= 0.5
    if atri or btri:
        fl *= 0.5
    return fl


def parse_shape(s):
    """GEMMLOG line split into words -> dict."""
    kv = dict(w.split("=") for w in s[4:] if "=" in w)
    return dict(M=int(s[1]), N=int(s[2]), K=int(s[3]), lower=int(kv["lower"]), atri=int(kv["atri"]),
                btri=int(kv["btri"]), inplace=int(kv.get("inplace", 0)), cfg=kv.get("cfg", "?"),
                part=kv.get("part", "main"), blocks=int(kv.get("blocks", 0)), cb=int(kv.get("cb", 0)))


def merge_parts(entries):
    """entries: list of (shape dict, duration seconds, extra) in launch order.  Returns logical launches:
    list of (shape dict of the main part, total duration, flops, number of kernels, extra of the main part)."""
    out = []
    for sh, t, extra in entries:
        if sh["part"] != "main" and out and (out[-1][0]["M"], out[-1][0]["N"], out[-1][0]["K"]) == (sh["M"], sh["N"], sh["K"]):
            m = out[-1]
            out[-1] = (m[0], m[1] + t, m[2], m[3] + 1, m[4])
        else:
            out.append((sh, t, launch_flops(sh["M"], sh["N"], sh["K"], sh["lower"], sh["atri"], sh["btri"], sh["cb"]), 1, extra))
    return out
