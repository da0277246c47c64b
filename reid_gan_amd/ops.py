"""Typed wrappers over the C ABI: torch tensors in, kernel launches on torch's current HIP stream.

PyTorch is used only as the owner of device memory and streams; every
computation below is a hand-written HIP kernel in libried_gan.so.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from ._lib import lib

BF16 = torch.bfloat16


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("reid_gan_amd runs on an MI355X (ROCm) device only; no GPU is visible and there is "
                           "no CPU fallback")
    lib.load()


def install_crash_backtrace(log_path: Optional[str] = None, stderr_fd: int = -1):
    """On a fatal signal, write the native frames of the faulting thread to stderr (or stderr_fd) and to log_path,
    then pass the signal on to the handler installed before (faulthandler).  Needs no GPU."""
    lib.ried_install_crash_backtrace(log_path.encode() if log_path else None, int(stderr_fd))


@dataclass
class View:
    """Channels [0, C) of an NHWC bf16 tensor: pixel p's channels start at ptr + 2*p*ld."""
    base: torch.Tensor          # keeps the storage alive
    ptr: int
    ld: int
    C: int
    N: int
    H: int
    W: int

    @property
    def npix(self) -> int:
        return self.N * self.H * self.W

    def slice(self, c0: int, c1: int) -> "View":
        assert 0 <= c0 < c1 <= self.C and c0 % 8 == 0 and c1 % 8 == 0, (c0, c1, self.C)
        return View(self.base, self.ptr + 2 * c0, self.ld, c1 - c0, self.N, self.H, self.W)

    def images(self, n0: int, n1: int) -> "View":
        assert 0 <= n0 < n1 <= self.N
        return View(self.base, self.ptr + 2 * n0 * self.H * self.W * self.ld, self.ld, self.C, n1 - n0, self.H, self.W)

    def torch(self) -> torch.Tensor:
        """The view as an (N,H,W,C) bf16 tensor (debug / tests)."""
        off = (self.ptr - self.base.data_ptr()) // 2
        flat = self.base.reshape(-1)
        return flat.as_strided((self.N, self.H, self.W, self.C),
                               (self.H * self.W * self.ld, self.W * self.ld, self.ld, 1), off)


def new_view(N: int, H: int, W: int, C: int, device="cuda", zero: bool = False) -> View:
    t = (torch.zeros if zero else torch.empty)((N, H, W, C), dtype=BF16, device=device)
    return View(t, t.data_ptr(), C, C, N, H, W)


# --------------------------------------------------------------------------- GEMM family
def conv2d(x: View, w: torch.Tensor, y: View, kh, kw, sh, sw, ph, pw, transposed=False, bias=None, act=0,
           stats=None, accumulate=False, out_f32=False, y2: Optional[torch.Tensor] = None,
           mask: Optional[View] = None, mask_slope=0.0, stats_post=False):
    """mask: view shaped like y; the stored value is multiplied by (mask > 0 ? 1 : mask_slope) (activation backward
    fused into a data gradient); stats_post: stats rows receive column sums of the stored values (True / 1), or
    the raw sums are added into the bn_fold_rows() rows of a zeroed buffer (2: the input of bn_apply_train)."""
    lib.ried_conv2d(x.ptr, x.ld, _p(w), y.ptr, y.ld, x.N, x.H, x.W, x.C, y.H, y.W, y.C, kh, kw, sh, sw, ph, pw,
                    int(transposed), _p(bias), act, _p(stats), int(accumulate), int(out_f32), _p(y2), y.C,
                    None if mask is None else mask.ptr, 0 if mask is None else mask.ld, mask_slope, int(stats_post),
                    _stream())


def conv2d_signs(x: View, w: torch.Tensor, y: View, kh, kw, sh, sw, ph, pw, signs: torch.Tensor, transposed=False,
                 bias=None, act=0, stats=None, accumulate=False, mask_slope=0.0, stats_post=False):
    """conv2d with the activation mask as sign bytes: signs uint8 [y pixels][y.C // 8] (conv_c1_fwd(..., signs=))."""
    assert signs.dtype == torch.uint8 and signs.is_contiguous() and signs.numel() == y.npix * (y.C // 8)
    lib.ried_conv2d_signs(x.ptr, x.ld, _p(w), y.ptr, y.ld, x.N, x.H, x.W, x.C, y.H, y.W, y.C, kh, kw, sh, sw, ph, pw,
                          int(transposed), _p(bias), act, _p(stats), int(accumulate), _p(signs), y.C // 8, mask_slope,
                          int(stats_post), _stream())


def conv2d_s2_dgrad_c1fold_scratch_floats(nimg, H, W) -> int:
    return lib.ried_conv2d_s2_dgrad_c1fold_scratch_floats(nimg, H, W)


def conv2d_s2_dgrad_c1fold(dy: View, wt: torch.Tensor, mask, mask_slope: float, stats: Optional[torch.Tensor],
                           ximg: torch.Tensor, dw: torch.Tensor, scratch: torch.Tensor, in_scale=1.0, in_shift=0.0):
    """Data gradient of a 3x3 stride-2 layer into the 64-channel output of a one-input-channel first layer, consumed
    in place: stats rows get the masked gradient's column sums, dw [64][9] += the first layer's weight gradient from
    its fp32 input ximg (nimg, 1, 2H, 2W); the gradient itself is not stored.  mask: the first layer's output (View)
    or its sign bytes (uint8 tensor [pixels][8])."""
    n, h, w = dy.N, dy.H, dy.W
    assert ximg.dtype == torch.float32 and ximg.is_contiguous() and ximg.numel() == n * 4 * h * w
    assert dw.numel() == 64 * 9 and scratch.numel() >= conv2d_s2_dgrad_c1fold_scratch_floats(n, h, w)
    if isinstance(mask, View):
        assert (mask.N, mask.H, mask.W, mask.C) == (n, 2 * h, 2 * w, 64)
        mp, ml, bits = mask.ptr, mask.ld, 0
    else:
        assert mask.dtype == torch.uint8 and mask.is_contiguous() and mask.numel() == n * 4 * h * w * 8
        mp, ml, bits = _p(mask), 8, 1
    lib.ried_conv2d_s2_dgrad_c1fold(dy.ptr, dy.ld, _p(wt), n, h, w, dy.C, mp, ml, bits, mask_slope, _p(stats), _p(ximg),
                                    in_scale, in_shift, _p(dw), _p(scratch), _stream())


def rows_add(part: torch.Tensor, nrows: int, width: int, out: torch.Tensor):
    """out[c] += sum over the first nrows rows of part[r][c]."""
    lib.ried_rows_add(_p(part), nrows, width, _p(out), _stream())


def conv2d_xf(x: View, w: torch.Tensor, y: View, kh, kw, sh, sw, ph, pw, xf_save, xf_gamma, xf_beta, xf_act=1,
              transposed=False, bias=None, act=0, stats=None, accumulate=False, stats_post=False):
    """conv2d on act(batchnorm(x)) with x the raw convolution output of the producing layer (save = its mean /
    invstd): the normalised tensor is formed inside the kernel, never stored."""
    lib.ried_conv2d_xf(x.ptr, x.ld, _p(w), y.ptr, y.ld, x.N, x.H, x.W, x.C, y.H, y.W, y.C, kh, kw, sh, sw, ph, pw,
                       int(transposed), _p(bias), act, _p(stats), int(accumulate), int(stats_post), _p(xf_save),
                       _p(xf_gamma), _p(xf_beta), int(xf_act), _stream())


def conv2d_which(x: View, w: torch.Tensor, y: View, kh, kw, sh, sw, ph, pw, transposed=False, bias=None, act=0,
                 stats=None, accumulate=False, out_f32=False, y2: Optional[torch.Tensor] = None,
                 mask: Optional[View] = None, mask_slope=0.0, stats_post=False) -> str:
    """Name of the kernel template a conv2d call with these arguments runs (for reports; nothing is launched)."""
    import ctypes
    which = (ctypes.c_int * 4)()
    lib.ried_conv2d_which(x.ptr, x.ld, _p(w), y.ptr, y.ld, x.N, x.H, x.W, x.C, y.H, y.W, y.C, kh, kw, sh, sw, ph, pw,
                          int(transposed), _p(bias), act, _p(stats), int(accumulate), int(out_f32), _p(y2), y.C,
                          None if mask is None else mask.ptr, 0 if mask is None else mask.ld, mask_slope,
                          int(stats_post), ctypes.addressof(which))
    kind, bm, bn, nst = list(which)
    if kind == 0 and nst >= 10:      # two K groups per block
        return f"conv_glds_kernel<{bm},{bn},{nst - 10},256,2>"
    return (f"conv_glds_kernel<{bm},{bn},{nst}>", f"conv_halo_kernel<{bm},{bn},{nst}>", f"conv_halo_s2t_kernel<{bn}>",
            f"conv_halo_d4_kernel<{bm}>", f"conv_halo_s2f_kernel<{bm}>")[kind]


def conv2d_tap_mse(x: View, w: torch.Tensor, bias: torch.Tensor, y: View, target: torch.Tensor, gout: View,
                   gscale: float, stats: torch.Tensor):
    """VGG tap layer on the prediction images with the perceptual MSE against `target` (fp32, same shape as y,
    contiguous) fused: stats rows get sum(d), sum(d*d) per channel, gout = gscale * d."""
    lib.ried_conv2d_tap_mse(x.ptr, x.ld, _p(w), _p(bias), y.ptr, y.ld, x.N, x.H, x.W, x.C, y.C, _p(target), y.C,
                            gout.ptr, gout.ld, gscale, _p(stats), _stream())


def conv2d_tap_mse_bf16(x: View, w: torch.Tensor, bias: torch.Tensor, y: View, target: View, gout: View,
                        gscale: float, stats: torch.Tensor):
    """conv2d_tap_mse against the target's stored bf16 activations (a view shaped like y)."""
    assert (target.N, target.H, target.W, target.C) == (y.N, y.H, y.W, y.C)
    lib.ried_conv2d_tap_mse_bf16(x.ptr, x.ld, _p(w), _p(bias), y.ptr, y.ld, x.N, x.H, x.W, x.C, y.C, target.ptr,
                                 target.ld, gout.ptr, gout.ld, gscale, _p(stats), _stream())


def stat_rows_total(stats: torch.Tensor, nrows: int, C: int, which: int, scale: float, out: torch.Tensor):
    """out[0] += scale * sum over rows and channels of stats[r][which][c]."""
    lib.ried_stat_rows_total(_p(stats), nrows, C, which, scale, _p(out), _stream())


def stat_rows_sum(stats: torch.Tensor, nrows: int, C: int, out: torch.Tensor):
    """out[c] += sum over the rows of stats[r][0][c]."""
    lib.ried_stat_rows_sum(_p(stats), nrows, C, _p(out), _stream())


def conv2d_wgrad(g: View, p: View, kh, kw, sh, sw, ph, pw, slab, splitk: int):
    """slab: fp32 tensor or raw device address of [splitk][kh*kw][p.C][g.C]."""
    lib.ried_conv2d_wgrad(g.ptr, g.ld, g.C, g.N, g.H, g.W, p.ptr, p.ld, p.C, p.H, p.W, kh, kw, sh, sw, ph, pw,
                          slab if isinstance(slab, int) else _p(slab), splitk, _stream())


class WgradJobs:
    """Weight gradients of many layers, launched together (one launch per tile shape)."""

    def __init__(self):
        self.groups = {}           # (bm, bn) -> [records], [block counts]
        self.dev = []
        self.flops = 0.0           # 2 * pixels * Cg * Cp * taps over all jobs (for reporting)
        self.group_flops, self.group_bytes = {}, {}

    def add(self, g: View, p: View, kh, kw, sh, sw, ph, pw, slab_ptr: int, splitk: int):
        import numpy as np
        rec = np.zeros(lib.ried_conv2d_wgrad_job_bytes(), dtype=np.uint8)
        tile, nb = np.zeros(2, dtype=np.int32), np.zeros(1, dtype=np.int32)
        lib.ried_conv2d_wgrad_make_job(rec.ctypes.data, g.ptr, g.ld, g.C, g.N, g.H, g.W, p.ptr, p.ld, p.C, p.H, p.W,
                                       kh, kw, sh, sw, ph, pw, slab_ptr, splitk, tile.ctypes.data, nb.ctypes.data)
        # (tile[1] carries the taps per block above bit 16: jobs of one launch must agree on that too)
        recs, counts = self.groups.setdefault((int(tile[0]), int(tile[1])), ([], []))
        recs.append(rec); counts.append((int(nb[0]), int(nb[0]) // splitk))       # (blocks, blocks per k split)
        self.flops += 2.0 * p.npix * p.C * g.C * kh * kw
        key = (int(tile[0]), int(tile[1]))
        self.group_flops[key] = self.group_flops.get(key, 0.0) + 2.0 * p.npix * p.C * g.C * kh * kw
        self.group_bytes[key] = (self.group_bytes.get(key, 0.0) + 2.0 * (g.npix * g.C + p.npix * p.C)
                                 + 4.0 * splitk * kh * kw * p.C * g.C)

    def finalize(self, device):
        import numpy as np
        self.dev = []
        for (bm, bn), (recs, counts) in sorted(self.groups.items()):
            jobs = torch.from_numpy(np.concatenate(recs)).to(device)
            m = self._xcd_order([(j, c, per) for j, (c, per) in enumerate(counts)])
            self.dev.append((bm, bn, jobs, torch.from_numpy(m).to(device), sum(c for c, _ in counts)))

    @staticmethod
    def _xcd_order(jobs):
        """map[pos] = (job, block of that job) for one launch.  Workgroups go to the 8 XCDs round-robin by their
        position; the blocks of a job that read the same pixels are those of one k split (`per` consecutive blocks),
        so a split's blocks are dealt to positions of one residue mod 8 as far as the counts allow (its pixels are
        then fetched into one L2 instead of eight).  Any permutation is valid: only the speed depends on it."""
        import numpy as np
        import os
        if os.environ.get("RIED_NO_XCD_MAP"):              # A/B switch: blocks in their own order
            return np.asarray([(j, b) for j, c, _ in jobs for b in range(c)], dtype=np.int32).reshape(-1)
        out, pos0 = [], 0
        for j, count, per in jobs:
            pos = np.arange(pos0, pos0 + count)
            blocks = np.arange(count)
            cls = (blocks // per) % 8                      # the XCD a split would like
            m = np.full(count, -1, dtype=np.int64)
            left_b, left_p = [], []
            for c in range(8):
                pc, bc = np.nonzero(pos % 8 == c)[0], blocks[cls == c]
                n = min(len(pc), len(bc))
                m[pc[:n]] = bc[:n]
                left_p.append(pc[n:]); left_b.append(bc[n:])
            lp, lb = np.concatenate(left_p), np.concatenate(left_b)
            m[lp] = lb
            assert sorted(m.tolist()) == list(range(count))
            out.append(np.stack([np.full(count, j, dtype=np.int64), m], axis=1))
            pos0 += count
        return np.concatenate(out).astype(np.int32).reshape(-1)

    def launch(self, around=None):
        """around(bm, bn, fn): reporting hook that runs fn (one tile shape's launch) between its own timers."""
        for bm, bn, jobs, m, n in self.dev:
            fn = lambda: lib.ried_conv2d_wgrad_jobs(_p(jobs), _p(m), n, bm, bn, _stream())
            if around is None:
                fn()
            else:
                around(bm, bn, fn)


class ConvJobs:
    """Convolutions that do not depend on each other, launched as one grid (ried_conv2d_jobs)."""

    def __init__(self):
        self.recs, self.counts = [], []
        self.jobs = None
        self.flops = self.bytes = 0.0      # 2 * pixels * Cin * Cout * taps, and each operand once (for reporting)

    def add(self, x: View, w: torch.Tensor, y: View, kh, kw, sh, sw, ph, pw, transposed=False, bias=None, act=0,
            stats=None, accumulate=False, stats_post=0):
        """Arguments as conv2d (statistics only in the folded form, stats_post = 2)."""
        import numpy as np
        assert self.jobs is None
        rec = np.zeros(lib.ried_conv2d_job_bytes(), dtype=np.uint8)
        nb = np.zeros(1, dtype=np.int32)
        lib.ried_conv2d_make_job(rec.ctypes.data, x.ptr, x.ld, _p(w), y.ptr, y.ld, x.N, x.H, x.W, x.C, y.H, y.W, y.C,
                                 kh, kw, sh, sw, ph, pw, int(transposed), _p(bias), act, _p(stats), int(accumulate),
                                 int(stats_post), nb.ctypes.data)
        self.recs.append(rec); self.counts.append(int(nb[0]))
        self.flops += 2.0 * (x.npix if transposed else y.npix) * x.C * y.C * kh * kw
        self.bytes += 2.0 * (x.npix * x.C + w.numel() + y.npix * y.C * (2 if accumulate else 1))

    def __len__(self):
        return len(self.recs)

    def finalize(self, device):
        import numpy as np
        self.jobs = torch.from_numpy(np.concatenate(self.recs)).to(device)
        m = np.asarray([(j, b) for j, c in enumerate(self.counts) for b in range(c)], dtype=np.int32).reshape(-1)
        self.map, self.nblocks = torch.from_numpy(m).to(device), sum(self.counts)

    def launch(self):
        lib.ried_conv2d_jobs(_p(self.jobs), _p(self.map), self.nblocks, _stream())


def conv2d_wgrad_tiles(Cg, Cp) -> int:
    return lib.ried_conv2d_wgrad_tiles(Cg, Cp)


def conv2d_wgrad_blocks(Cg, Cp, kh, kw, npix) -> int:
    """Blocks per k split of conv2d_wgrad over npix output pixels: tiles x taps / the taps one block computes."""
    return lib.ried_conv2d_wgrad_tiles(Cg, Cp) * (kh * kw // lib.ried_conv2d_wgrad_taps_per_block(Cg, Cp, kh, kw, npix))


def wgrad_reduce(slab, splitk, taps, Cp, Cg, out: torch.Tensor, st, sp, sg, scale=1.0, accumulate=False):
    lib.ried_wgrad_reduce(_p(slab), splitk, taps, Cp, Cg, _p(out), st, sp, sg, scale, int(accumulate), _stream())


def attn_scores_softmax_supported(n: int, d: int) -> bool:
    return bool(lib.ried_attn_scores_softmax_supported(n, d))


def attn_scores_softmax(qkv: View, d: int, P: torch.Tensor):
    """P[b][i][:] = softmax_j(q_i . k_j): q = channels [0, d), k = [d, 2d) of qkv; one pass, no fp32 energies."""
    lib.ried_attn_scores_softmax(qkv.ptr, qkv.ld, qkv.N, qkv.H * qkv.W, d, _p(P), _stream())


def attn_fused_supported(n: int, d: int, C: int) -> bool:
    return bool(lib.ried_attn_fused_supported(n, d, C))


def attn_fwd(qkv: View, x: View, out: View, O: View, lse: torch.Tensor, gamma: torch.Tensor, d: int):
    """Fused Self_Attn forward on the [q; k; v] projection: O = softmax(q^T k) v, out = gamma * O + x, lse per query."""
    lib.ried_attn_fwd(qkv.ptr, qkv.ld, x.ptr, x.ld, out.ptr, out.ld, O.ptr, O.ld, _p(lse), _p(gamma), qkv.N,
                      qkv.H * qkv.W, d, O.C, _stream())


def attn_bwd(qkv: View, dqkv: View, dout: View, O: View, lse_ptr: int, gamma: torch.Tensor, dgamma, d: int):
    """Fused Self_Attn backward: dqkv = gradients of q, k, v; dgamma[0] += sum dout * O (None: skipped)."""
    lib.ried_attn_bwd(qkv.ptr, qkv.ld, dqkv.ptr, dqkv.ld, dout.ptr, dout.ld, O.ptr, O.ld, lse_ptr, _p(gamma),
                      _p(dgamma), qkv.N, qkv.H * qkv.W, d, O.C, _stream())


def bgemm(A, lda, bsa, a_tr, B, ldb, bsb, b_tr, M, N, K, batch, out, ldc, bsc, out_f32=False, alpha=1.0,
          gamma=None, resid=None, ld_resid=0, bs_resid=0, accumulate=False):
    lib.ried_bgemm(A, lda, bsa, int(a_tr), B, ldb, bsb, int(b_tr), M, N, K, batch, out, ldc, bsc, int(out_f32),
                   alpha, _p(gamma), resid, ld_resid, bs_resid, int(accumulate), _stream())


def pack_weight(src: torch.Tensor, dst: torch.Tensor, N, R, S, C, sn, sr, ss, sc, sigma=None):
    lib.ried_pack_weight(_p(src), _p(dst), N, R, S, C, sn, sr, ss, sc, _p(sigma), _stream())


JOB_CHUNK = 2048


class JobTable:
    """Host-side builder of a device job table for ried_pack_jobs / ried_wgrad_reduce_jobs: rows of int64 words
    (pointers and integers) plus the (job, chunk) map that assigns 2048-element chunks to blocks."""

    def __init__(self, words: int):
        self.words, self.rows, self.totals = words, [], []
        self.jobs = self.blockmap = None

    def add(self, row, total: int = 0, nblocks: Optional[int] = None) -> int:
        """total: elements (one block per 2048); or nblocks: the job's own block count (blockmap[b][1] = 0..n-1)."""
        assert len(row) == self.words and self.jobs is None
        self.rows.append([int(v) for v in row])
        self.totals.append(int(nblocks) if nblocks is not None else (int(total) + JOB_CHUNK - 1) // JOB_CHUNK)
        return len(self.rows) - 1

    def finalize(self, device):
        import numpy as np
        if not self.rows:
            return
        bm = []
        for j, t in enumerate(self.totals):
            bm.extend((j, c) for c in range(t))
        self.jobs = torch.from_numpy(np.asarray(self.rows, dtype=np.int64)).to(device)
        self.blockmap = torch.from_numpy(np.asarray(bm, dtype=np.int32).reshape(-1)).to(device)
        self.nblocks = len(bm)


def pack_jobs(t: JobTable):
    if t.jobs is not None:
        lib.ried_pack_jobs(_p(t.jobs), _p(t.blockmap), t.nblocks, _stream())


def wgrad_reduce_jobs(t: JobTable):
    if t.jobs is not None:
        lib.ried_wgrad_reduce_jobs(_p(t.jobs), _p(t.blockmap), t.nblocks, _stream())


class SnJobs:
    """All spectral-norm layers of a network: one job row per layer and the block maps of the five kernels."""

    def __init__(self):
        self.rows, self.keep = [], []
        self.jobs = None

    def add(self, W, u, v, sigma, R, K, dWsn=None, dW=None):
        dev = W.device
        t = torch.empty(max(K, 64), dtype=torch.float32, device=dev)
        s = torch.empty(R, dtype=torch.float32, device=dev)
        self.keep += [t, s]
        self.rows.append([W.data_ptr(), u.data_ptr(), v.data_ptr(), sigma.data_ptr(), t.data_ptr(), s.data_ptr(), R, K,
                          0 if dWsn is None else dWsn.data_ptr(), 0 if dW is None else dW.data_ptr()])

    def finalize(self, device):
        import numpy as np
        if not self.rows:
            return
        def bmap(count):
            m = [(j, c) for j, r in enumerate(self.rows) for c in range(count(r[6], r[7]))]
            return torch.from_numpy(np.asarray(m, dtype=np.int32).reshape(-1)).to(device), len(m)
        self.jobs = torch.from_numpy(np.asarray(self.rows, dtype=np.int64)).to(device)
        self.map_cols, self.n_cols = bmap(lambda R, K: (K + 63) // 64)
        self.map_rows, self.n_rows = bmap(lambda R, K: (R + 3) // 4)
        self.map_dot, self.n_dot = bmap(lambda R, K: 64)
        self.map_apply, self.n_apply = bmap(lambda R, K: (R * K + 2047) // 2048)


def sn_sigma_jobs(j: SnJobs, power_iter=True, eps=1e-12):
    lib.ried_sn_sigma_jobs(_p(j.jobs), len(j.rows), _p(j.map_cols), j.n_cols, _p(j.map_rows), j.n_rows,
                           int(power_iter), eps, _stream())


def sn_backward_jobs(j: SnJobs, accumulate=False):
    lib.ried_sn_backward_jobs(_p(j.jobs), _p(j.map_dot), j.n_dot, _p(j.map_apply), j.n_apply, int(accumulate),
                              _stream())


# --------------------------------------------------------------------------- batch norm
def conv2d_stat_rows(nimg, Ho, Wo, Cin, Cout, kh, kw, sh=1, sw=1, transposed=False) -> int:
    """Rows of the stats buffer a conv2d call with these dimensions (Ho, Wo: its output) writes."""
    return lib.ried_conv2d_stat_rows(nimg, Ho, Wo, Cin, Cout, kh, kw, sh, sw, int(transposed))


def conv_c1_stat_rows(nimg, H, W, Cout, stride) -> int:
    return lib.ried_conv_c1_stat_rows(nimg, H, W, Cout, stride)


def bn_backward_scratch_floats(C) -> int:
    return lib.ried_bn_backward_scratch_floats(C)


def colsum_scratch_floats(C) -> int:
    return lib.ried_colsum_scratch_floats(C)


def bn_finalize_fold_floats(C) -> int:
    return lib.ried_bn_finalize_fold_floats(C)


def bn_finalize(stats, nrows, C, count, save, rm=None, rv=None, nbt=None, eps=1e-5, momentum=0.1, fold=None,
                ldstat: int = 0):
    """ldstat: channels per statistics row when the layer is a slice of a wider convolution (0: C)."""
    lib.ried_bn_finalize(_p(stats), nrows, C, ldstat, count, eps, momentum, _p(save), _p(rm), _p(rv), _p(nbt),
                         _p(fold), _stream())


def bn_eval_stats(rm, rv, C, save, eps=1e-5):
    lib.ried_bn_eval_stats(_p(rm), _p(rv), C, eps, _p(save), _stream())


ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2, 3     # the `relu` argument of the bn_* calls takes these too


def bn_apply(x: View, y: View, save, gamma, beta, relu=True, accumulate=False):
    lib.ried_bn_apply(x.ptr, x.ld, y.ptr, y.ld, x.npix, x.C, _p(save), _p(gamma), _p(beta), int(relu),
                      int(accumulate), _stream())


def bn_finalize_images(stats, rows_per_img: int, nimg: int, C: int, count: int, save, eps=1e-5, ldstat: int = 0):
    """Per-image (mean, invstd) [nimg][2][C] from partial rows [nimg * rows_per_img][2][C] (ldstat: channels per row
    when the layer is a channel slice of a wider convolution's rows)."""
    lib.ried_bn_finalize_images(_p(stats), rows_per_img, nimg, C, ldstat, count, eps, _p(save), _stream())


def bn_apply_images(x: View, y: View, save, gamma, beta, relu=True, accumulate=False):
    """bn_apply with one statistics set per image of the views."""
    lib.ried_bn_apply_images(x.ptr, x.ld, y.ptr, y.ld, x.H * x.W, x.N, x.C, _p(save), _p(gamma), _p(beta), int(relu),
                             int(accumulate), _stream())


def bn_backward(dy: View, x: View, dx: View, save, gamma, beta, scratch, dgamma=None, dbeta=None, relu=True,
                accumulate=False, counter: Optional[torch.Tensor] = None, folded: Optional[torch.Tensor] = None):
    """scratch: fp32, at least bn_backward_scratch_floats(C) elements.  counter: a zeroed int32 element owned by
    the layer; with it the reduction's last block finishes the sums (one launch fewer).  folded: zeroed fp32
    [bn_backward_folded_floats(C)] owned by the layer (left non-zero): two launches, atomically folded sums.
    folded and counter (a zeroed word, left non-zero): one launch with a grid barrier, bn_backward_fused_ok() layers."""
    lib.ried_bn_backward(dy.ptr, dy.ld, x.ptr, x.ld, dx.ptr, dx.ld, x.npix, x.C, _p(save), _p(gamma), _p(beta),
                         int(relu), _p(scratch), _p(dgamma), _p(dbeta), int(accumulate), _p(counter), _p(folded),
                         _stream())


def bn_backward_reduce(dy: View, x: View, save, gamma, beta, folded: torch.Tensor, relu=True):
    """First launch of the folded BatchNorm backward: folded (zeroed) += sums of g and g * x-hat."""
    lib.ried_bn_backward_reduce(dy.ptr, dy.ld, x.ptr, x.ld, x.npix, x.C, _p(save), _p(gamma), _p(beta), int(relu),
                                _p(folded), _stream())


def bn_backward_apply(dy: View, x: View, dx: View, save, gamma, beta, folded: torch.Tensor, dgamma=None, dbeta=None,
                      relu=True, accumulate=False):
    """Second launch: dx (and the parameter gradients) from the rows of `folded`."""
    lib.ried_bn_backward_apply(dy.ptr, dy.ld, x.ptr, x.ld, dx.ptr, dx.ld, x.npix, x.C, _p(save), _p(gamma), _p(beta),
                               int(relu), _p(folded), _p(dgamma), _p(dbeta), int(accumulate), _stream())


def bn_backward_folded_floats(C: int) -> int:
    return lib.ried_bn_backward_folded_floats(C)


def bn_backward_fused_ok(npix: int, maxC: int, njobs: int = 1) -> bool:
    """Whether njobs layers of npix pixels and at most maxC channels are small enough for the one-launch backward."""
    return bool(lib.ried_bn_backward_fused_ok(npix, maxC, njobs))


def barrier_timeouts() -> int:
    """Blocks that gave up waiting at a grid barrier since the library was loaded (0 in a healthy process)."""
    import ctypes
    n = ctypes.c_uint(0)
    lib.ried_barrier_timeouts(ctypes.addressof(n))
    return n.value


def bn_backward_job(dy: View, x: View, dx: View, save, gamma, beta, scratch, dgamma, dbeta, act=1, accumulate=False,
                    folded: Optional[torch.Tensor] = None):
    """One 16-word record of the table bn_backward_jobs takes."""
    return [dy.ptr, dy.ld, x.ptr, x.ld, dx.ptr, dx.ld, x.C, _p(save), _p(gamma), _p(beta), int(act),
            _p(scratch) or 0, _p(dgamma) or 0, _p(dbeta) or 0, int(accumulate), _p(folded) or 0]


def bn_backward_jobs(table: torch.Tensor, njobs: int, npix: int, maxC: int, all_folded: int = 0):
    """table: int64 device tensor [njobs][16] of bn_backward_job records (layers sharing the pixel count).
    all_folded: 1 every job has folded rows (two launches); 2 also a zeroed barrier counter in its scratch word and
    bn_backward_fused_ok(npix, maxC, njobs) holds (one launch)."""
    lib.ried_bn_backward_jobs(_p(table), njobs, npix, maxC, int(all_folded), _stream())


def bn_apply_train_job(x: View, y: View, stats_ptr: int, nrows, ldstat, save, rm, rv, nbt, gamma, beta, act=1,
                       accumulate=False):
    """One 16-word record of the table bn_apply_train_jobs takes."""
    return [x.ptr, x.ld, y.ptr, y.ld, x.C, stats_ptr, nrows, ldstat, _p(save), _p(rm), _p(rv), _p(nbt), _p(gamma),
            _p(beta), int(act), int(accumulate)]


def bn_apply_train_jobs(table: torch.Tensor, njobs: int, npix: int, maxC: int, minC: int, eps=1e-5, momentum=0.1):
    lib.ried_bn_apply_train_jobs(_p(table), njobs, npix, maxC, minC, eps, momentum, _stream())


def bn_fold_rows() -> int:
    return lib.ried_bn_fold_rows()


def bn_apply_train(x: View, y: View, stats, nrows, save, gamma, beta, rm=None, rv=None, nbt=None, relu=True,
                   accumulate=False, eps=1e-5, momentum=0.1, ldstat: int = 0):
    """bn_finalize + bn_apply in one launch, from the folded rows conv2d(..., stats_post=2) accumulated.
    ldstat: channels per statistics row when the layer is a slice of a wider convolution (0: x.C)."""
    lib.ried_bn_apply_train(x.ptr, x.ld, y.ptr, y.ld, x.npix, x.C, _p(stats), nrows, ldstat, eps, momentum, _p(save),
                            _p(rm), _p(rv), _p(nbt), _p(gamma), _p(beta), int(relu), int(accumulate), _stream())


def conv2d_bn(x: View, w: torch.Tensor, y: View, kh, kw, sh, sw, ph, pw, stats, counter, save, rm=None, rv=None,
              nbt=None, transposed=False, eps=1e-5, momentum=0.1):
    """Convolution + train-mode BatchNorm statistics in one launch (y = raw conv result, save = mean / invstd)."""
    lib.ried_conv2d_bn(x.ptr, x.ld, _p(w), y.ptr, y.ld, x.N, x.H, x.W, x.C, y.H, y.W, y.C, kh, kw, sh, sw, ph, pw,
                       int(transposed), _p(stats), _p(counter), eps, momentum, _p(save), _p(rm), _p(rv), _p(nbt),
                       _stream())


# --------------------------------------------------------------------------- pooling etc.
POOL_MAX3S2, POOL_MAX2S2, POOL_AVG3 = 0, 1, 2


def pool_fwd(x: View, y: View, mode: int):
    lib.ried_pool_fwd(x.ptr, x.ld, y.ptr, y.ld, x.N, x.H, x.W, x.C, mode, _stream())


def avgpool3_stats_rows(npix: int, C: int) -> int:
    return lib.ried_avgpool3_stats_rows(npix, C)


def avgpool3_stats(x: View, y: View, stats: torch.Tensor):
    """y = avg_pool2d(x, 3, 1, 1); stats[rows][2][C] = per-block column sums of y and y * y (for bn_finalize)."""
    lib.ried_avgpool3_stats(x.ptr, x.ld, y.ptr, y.ld, x.N, x.H, x.W, x.C, _p(stats), _stream())


def pool_bwd(x: Optional[View], dy: View, dx: View, mode: int, accumulate=True, relu_mask=False):
    lib.ried_pool_bwd(None if x is None else x.ptr, 0 if x is None else x.ld, dy.ptr, dy.ld, dx.ptr, dx.ld, dx.N,
                      dx.H, dx.W, dx.C, mode, int(accumulate), int(relu_mask), _stream())


def upsample2x_fwd(x: View, y: View):
    lib.ried_upsample2x_fwd(x.ptr, x.ld, y.ptr, y.ld, x.N, x.H, x.W, x.C, _stream())


def upsample2x_bwd(dy: View, dx: View):
    lib.ried_upsample2x_bwd(dy.ptr, dy.ld, dx.ptr, dx.ld, dx.N, dx.H, dx.W, dx.C, _stream())


EW_COPY, EW_ADD, EW_ACTBWD, EW_ACTBWD_ADD, EW_AXPY, EW_SCALE = range(6)


def ew(op: int, a: View, out: View, y: Optional[View] = None, slope: float = 0.0, s=None):
    lib.ried_ew(op, a.ptr, a.ld, None if y is None else y.ptr, 0 if y is None else y.ld, out.ptr, out.ld, a.npix,
                a.C, slope, _p(s), _stream())


def colsum(a: View, out: torch.Tensor, scratch: torch.Tensor):
    """out[c] += sum over pixels; scratch: fp32, at least colsum_scratch_floats(C) elements."""
    lib.ried_colsum(a.ptr, a.ld, a.npix, a.C, _p(out), _p(scratch), _stream())


def dot(a: View, b: Optional[View], scale: float, out: torch.Tensor):
    lib.ried_dot(a.ptr, a.ld, None if b is None else b.ptr, 0 if b is None else b.ld, a.npix, a.C, scale, _p(out),
                 _stream())


def fill(out: View, value: float, s=None):
    lib.ried_fill(out.ptr, out.ld, out.npix, out.C, _p(s), value, _stream())


def softmax_fwd(S: torch.Tensor, P: torch.Tensor, rows: int, n: int):
    lib.ried_softmax_fwd(_p(S), _p(P), rows, n, _stream())


def softmax_bwd(dP: torch.Tensor, P: torch.Tensor, dS: torch.Tensor, rows: int, n: int):
    lib.ried_softmax_bwd(_p(dP), _p(P), _p(dS), rows, n, _stream())


def nchw_to_nhwc(x: torch.Tensor, y: View):
    lib.ried_nchw_to_nhwc(_p(x), y.ptr, y.ld, y.N, y.C, y.H * y.W, _stream())


def nhwc_to_nchw(x: View, y: torch.Tensor):
    lib.ried_nhwc_to_nchw(x.ptr, x.ld, _p(y), x.N, x.C, x.H * x.W, _stream())


def adam(p, g, m, v, lr, step: torch.Tensor, b1=0.01, b2=0.99, eps=1e-8):
    """step: int32 device tensor (1,), number of completed steps; incremented by the call."""
    lib.ried_adam(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, b1, b2, eps, _p(step), _stream())


def sum_f32(x: torch.Tensor, n: int, stride: int, scale: float, out: torch.Tensor, square: bool = False):
    lib.ried_sum_f32(_p(x), n, stride, scale, int(square), _p(out), _stream())


# --------------------------------------------------------------------------- special layers
def conv_c1_fwd(x: torch.Tensor, w, y: View, stride, sigma=None, bias=None, in_scale=1.0, in_shift=0.0, act=0,
                stats=None, signs: Optional[torch.Tensor] = None):
    """signs: optional uint8 [pixels of y][y.C // 8] that receives one bit per stored value (positive or not)."""
    n, h, wd = x.shape[0], x.shape[-2], x.shape[-1]
    if signs is not None:
        assert signs.dtype == torch.uint8 and signs.is_contiguous() and signs.numel() == y.npix * (y.C // 8)
        lib.ried_conv_c1_fwd_signs(_p(x), _p(w), _p(sigma), _p(bias), y.ptr, y.ld, n, h, wd, y.C, stride, in_scale,
                                   in_shift, act, _p(stats), _p(signs), _stream())
        return
    lib.ried_conv_c1_fwd(_p(x), _p(w), _p(sigma), _p(bias), y.ptr, y.ld, n, h, wd, y.C, stride, in_scale, in_shift,
                         act, _p(stats), _stream())


def conv_c1_wgrad_scratch_floats(Cout) -> int:
    return lib.ried_conv_c1_wgrad_scratch_floats(Cout)


def conv_c1_wgrad(x: torch.Tensor, dy: View, stride, dw: torch.Tensor, scratch: torch.Tensor, in_scale=1.0,
                  in_shift=0.0):
    """dw += weight gradient; scratch: fp32, at least conv_c1_wgrad_scratch_floats(Cout) elements."""
    n, h, wd = x.shape[0], x.shape[-2], x.shape[-1]
    lib.ried_conv_c1_wgrad(_p(x), dy.ptr, dy.ld, n, h, wd, dy.C, stride, in_scale, in_shift, _p(dw), _p(scratch),
                           _stream())


def conv_c1_dgrad(dy: View, w, dx: torch.Tensor, sigma=None, scale=1.0, accumulate=False):
    lib.ried_conv_c1_dgrad(dy.ptr, dy.ld, _p(w), _p(sigma), _p(dx), dy.N, dy.H, dy.W, dy.C, scale, int(accumulate),
                           _stream())


def conv_c1_dgrad_s2(dy: View, w, dx: torch.Tensor, sigma=None, scale=1.0, accumulate=False):
    """dx: fp32 (N,1,H,W) image gradient of the stride-2 one-channel conv whose output gradient is dy."""
    lib.ried_conv_c1_dgrad_s2(dy.ptr, dy.ld, _p(w), _p(sigma), _p(dx), dx.shape[0], dx.shape[-2], dx.shape[-1], dy.C,
                              scale, int(accumulate), _stream())


def conv3_to1_fwd(x: View, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor):
    """out fp32 (N,1,H,W) = Conv2d(C, 1, 3, padding=1)(x) with the fp32 master weight (1,C,3,3)."""
    lib.ried_conv3_to1_fwd(x.ptr, x.ld, _p(w), _p(bias), _p(out), x.N, x.H, x.W, x.C, _stream())


def conv3_to1_dgrad(dout: torch.Tensor, w: torch.Tensor, dx: View, accumulate=False):
    lib.ried_conv3_to1_dgrad(_p(dout), _p(w), dx.ptr, dx.ld, dx.N, dx.H, dx.W, dx.C, int(accumulate), _stream())


def conv3_to1_wgrad(x: View, dout: torch.Tensor, dw: torch.Tensor):
    """dw (1,C,3,3) += weight gradient."""
    lib.ried_conv3_to1_wgrad(x.ptr, x.ld, _p(dout), _p(dw), x.N, x.H, x.W, x.C, _stream())


def axpy_f32(y: torch.Tensor, x: torch.Tensor, alpha: float):
    lib.ried_axpy_f32(_p(y), _p(x), alpha, y.numel(), _stream())


def out_conv_fwd(x: torch.Tensor, h: View, w: torch.Tensor, out: torch.Tensor, bn=None, bn_images: int = 0):
    """bn = (save, gamma, beta): h is the raw conv output and BatchNorm + ReLU are applied on the fly;
    bn_images > 1: save holds one statistics set per image."""
    save, gamma, beta = bn if bn is not None else (None, None, None)
    lib.ried_out_conv_fwd(_p(x), h.ptr, h.ld, _p(w), _p(out), h.npix, _p(save), _p(gamma), _p(beta), bn_images,
                          _stream())


def out_conv_bwd(x, h: View, w, dout, dh: View, dw):
    lib.ried_out_conv_bwd(_p(x), h.ptr, h.ld, _p(w), _p(dout), dh.ptr, dh.ld, _p(dw), h.npix, _stream())


def sn_sigma(Wm: torch.Tensor, R, K, u, v, sigma, scratch, power_iter=True, eps=1e-12):
    lib.ried_sn_sigma(_p(Wm), R, K, _p(u), _p(v), _p(sigma), _p(scratch), int(power_iter), eps, _stream())


def sn_backward(dWsn, Wm, u, v, sigma, R, K, dW, scratch1, accumulate=False):
    lib.ried_sn_backward(_p(dWsn), _p(Wm), _p(u), _p(v), _p(sigma), R, K, _p(dW), _p(scratch1), int(accumulate),
                         _stream())


def l2norm_tap(a_ptr, b_ptr, npix, C, loss_scale, loss, g_ptr, grad_scale):
    """loss[0] += loss_scale * sum over pixels of the channel L2 norm of (a - b); g = grad_scale * d/da."""
    lib.ried_l2norm_tap(a_ptr, b_ptr, npix, C, loss_scale, _p(loss), g_ptr, grad_scale, _stream())


def mse_tap(a_ptr, b_ptr, n, loss_scale, loss, g_ptr, grad_scale, accumulate=False):
    lib.ried_mse_tap(a_ptr, b_ptr, n, loss_scale, _p(loss), g_ptr, grad_scale, int(accumulate), _stream())


def dloss_scalars(sc: torch.Tensor, n_per_half: int, keep: Optional[torch.Tensor] = None):
    lib.ried_dloss_scalars(_p(sc), n_per_half, _p(keep) if keep is not None else None, _stream())


def gloss_scalars(sc: torch.Tensor, wv, ws, wg):
    lib.ried_gloss_scalars(_p(sc), wv, ws, wg, _stream())


def ssim_fwd(a, b, ref, c1, c2, sums, gmap=None, clamp=None):
    n, h, w = ref.shape[0], ref.shape[-2], ref.shape[-1]
    lo, hi = clamp if clamp is not None else (0.0, 0.0)
    lib.ried_ssim_fwd(_p(a), _p(b), _p(ref), n, h, w, c1, c2, int(clamp is not None), lo, hi, _p(sums), _p(gmap),
                      _stream())


def ssim_bwd(a, ref, gmap, scale, da, scale_dev=None, accumulate=False):
    n, h, w = ref.shape[0], ref.shape[-2], ref.shape[-1]
    lib.ried_ssim_bwd(_p(a), _p(ref), _p(gmap), n, h, w, scale, _p(scale_dev), _p(da), int(accumulate), _stream())
