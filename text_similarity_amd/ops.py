"""Device ops: thin wrappers that pass raw device pointers of torch tensors to libtsim.so.

torch is used for allocation, stream identity and host<->device copies only.  Every function requires
CUDA (ROCm) tensors and raises otherwise: there is no CPU path in the product."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib


def _need_gpu(*ts):
    dev = None
    for t in ts:
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise _lib.TsimError("text_similarity_amd ops run on MI355X only: expected a CUDA/ROCm tensor, "
                                 f"got {type(t).__name__} on {getattr(t, 'device', None)}")
        if dev is not None and t.device != dev:
            raise ValueError(f"operands on different devices: {dev} and {t.device}")
        dev = t.device


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


UNIT_DTYPE = torch.float16      # storage type of the unit rows the MFMA search kernel streams (csrc/common.h unit_t)

_workspaces = {}


def _workspace(dev: torch.device, nbytes: int) -> torch.Tensor:
    """Scratch for one search call, keyed by (device, current stream): calls on different streams never share it, and
    calls on one stream are ordered by the stream."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    w = _workspaces.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _workspaces[key] = w
    return w


def pad_dim(d: int) -> int:
    p = _lib.lib().tsim_pad_dim(int(d))
    if p == 0:
        raise ValueError(f"embedding width {d} > 768 is not supported by the search kernels")
    return p


def new_rho(device) -> torch.Tensor:
    """A zeroed device float for the rounding-residual maximum of a set of unit rows (see :func:`l2norm_rows`)."""
    return torch.zeros((1,), dtype=torch.float32, device=device)


def l2norm_rows(x: torch.Tensor, eps: float = 1e-8, rho: Optional[torch.Tensor] = None, return_rho: bool = False):
    """[rows, d] float32/bf16 -> unit rows in float16 (IEEE half), zero-padded to [rows, pad_dim(d)] (A7 operand prep).

    ``rho`` (a device float32 tensor of one element, e.g. from :func:`new_rho`) is atomically raised to the largest
    rounding residual ||half(u_r) - u_r||_2 of the rows written; several calls may accumulate into one word (a corpus built
    chunk by chunk).  ``return_rho=True`` allocates a fresh word and returns ``(unit_rows, rho)``.  Passing that word to
    :func:`cosine_topk` as ``rho_c`` gives the search's exactness guard its measured (tightest) error bound."""
    _need_gpu(x)
    if x.dim() != 2:
        raise ValueError("l2norm_rows expects a 2-D tensor")
    if x.dtype not in (torch.float32, torch.bfloat16):
        x = x.float()
    x = x.contiguous()
    rows, d = x.shape
    ld = pad_dim(d)
    out = torch.empty((rows, ld), dtype=UNIT_DTYPE, device=x.device)
    dt = _lib.TSIM_F32 if x.dtype == torch.float32 else _lib.TSIM_BF16
    if rho is None and return_rho:
        rho = new_rho(x.device)
    if rho is not None:
        _check_rho(rho, x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().tsim_l2norm_rows(x.data_ptr(), dt, rows, d, x.stride(0), out.data_ptr(), ld, eps,
                                               rho.data_ptr() if rho is not None else 0, _stream(x)), "l2norm_rows")
    return (out, rho) if return_rho else out


def _check_rho(rho, dev):
    if not isinstance(rho, torch.Tensor) or rho.dtype != torch.float32 or rho.numel() != 1 or rho.device != dev:
        raise ValueError(f"rho must be a float32 tensor of one element on {dev}")


MAX_QUERIES_PER_CALL = 16384    # workspace grows by ~16 KB + 512 k bytes per query: larger query sets are searched in slices


def cosine_topk(eq_unit: torch.Tensor, ec_unit: torch.Tensor, d: int, k: int, idx_offset: int = 0,
                eq_f32: Optional[torch.Tensor] = None, ec_f32: Optional[torch.Tensor] = None,
                return_status: bool = False, rho_c: Optional[torch.Tensor] = None,
                out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """Top-k of every query row against every corpus row: scores [Q,k] f32, idx [Q,k] i64, ordered by (score desc,
    index asc).  ``eq_unit`` / ``ec_unit`` are the unit float16 rows from :func:`l2norm_rows` (what the MFMA kernel streams).
    With ``eq_f32`` / ``ec_f32`` (the float32 embeddings the unit rows were made from) the returned scores are the
    reference's ``F.cosine_similarity`` of the float32 rows (/root/reference/src/pipeline/search_pipeline.py:76-78) and the
    order is exact for them; without, the inner product of the unit rows as stored.  ``rho_c``: the residual maximum of
    ``ec_unit`` from :func:`l2norm_rows` (tightens the guard's proven error bound; without it the a-priori bound of a
    correctly rounded unit row is used — results are exact either way, more queries take the widening pass).
    ``return_status`` adds an int32 [Q] tensor: 0 = first pass, 1 = widened, 2 = brute force (include/tsim.h).
    1 <= k <= 64, d <= 768.  Query sets above MAX_QUERIES_PER_CALL rows are searched in slices (queries are independent).
    ``out`` = (scores, idx): preallocated contiguous [Q,k] float32 / int64 tensors to write into (e.g. two views of one
    exchange buffer, :func:`packed_result_buffer`)."""
    _need_gpu(eq_unit, ec_unit)
    if eq_unit.dtype != UNIT_DTYPE or ec_unit.dtype != UNIT_DTYPE:
        raise ValueError("cosine_topk expects float16 unit rows from l2norm_rows")
    ld = pad_dim(d)
    if eq_unit.shape[1] != ld or ec_unit.shape[1] != ld or not eq_unit.is_contiguous() or not ec_unit.is_contiguous():
        raise ValueError(f"cosine_topk: rows must be contiguous with stride pad_dim({d})={ld}")
    if (eq_f32 is None) != (ec_f32 is None):
        raise ValueError("cosine_topk: pass both float32 matrices or neither")
    Q, N = eq_unit.shape[0], ec_unit.shape[0]
    dev = eq_unit.device
    if ec_unit.device != dev:
        raise ValueError(f"cosine_topk: operands on different devices ({dev} vs {ec_unit.device})")
    qf = cf = 0
    ldq = ldc = 0
    if eq_f32 is not None:
        _need_gpu(eq_f32, ec_f32)
        for t, rows, name in ((eq_f32, Q, "eq_f32"), (ec_f32, N, "ec_f32")):
            if t.dtype != torch.float32 or t.dim() != 2 or t.shape != (rows, d) or t.stride(1) != 1 or t.device != dev:
                raise ValueError(f"cosine_topk: {name} must be float32 [{rows}, {d}] with unit inner stride on {dev}")
        qf, cf, ldq, ldc = eq_f32.data_ptr(), ec_f32.data_ptr(), eq_f32.stride(0), ec_f32.stride(0)
    if rho_c is not None:
        _check_rho(rho_c, dev)
    if out is not None:
        scores, idx = out
        _need_gpu(scores, idx)
        if (scores.shape != (Q, k) or idx.shape != (Q, k) or scores.dtype != torch.float32 or idx.dtype != torch.int64
                or not scores.is_contiguous() or not idx.is_contiguous() or scores.device != dev or idx.device != dev):
            raise ValueError(f"cosine_topk: out must be contiguous float32 / int64 [{Q}, {k}] tensors on {dev}")
    else:
        scores = torch.empty((Q, k), dtype=torch.float32, device=dev)
        idx = torch.empty((Q, k), dtype=torch.int64, device=dev)
    status = torch.zeros((Q,), dtype=torch.int32, device=dev) if return_status else None
    if Q == 0:
        return (scores, idx, status) if return_status else (scores, idx)
    L = _lib.lib()
    with torch.cuda.device(dev):
        step = min(Q, MAX_QUERIES_PER_CALL)
        nbytes = L.tsim_cosine_topk_workspace_bytes(step, N, k)
        if nbytes == 0:
            raise ValueError(f"cosine_topk: unsupported shape Q={Q} N={N} k={k} (1 <= k <= 64)")
        ws = _workspace(dev, nbytes)
        for q0 in range(0, Q, step):
            nq = min(step, Q - q0)
            _lib.check(L.tsim_cosine_topk_ex(eq_unit.data_ptr() + q0 * ld * 2, qf + q0 * ldq * 4 if qf else 0, ldq, nq,
                                             ec_unit.data_ptr(), cf, ldc, rho_c.data_ptr() if rho_c is not None else 0, N, d, ld, k,
                                             scores.data_ptr() + q0 * k * 4, idx.data_ptr() + q0 * k * 8,
                                             status.data_ptr() + q0 * 4 if return_status else 0,
                                             idx_offset, ws.data_ptr(), ws.numel(), _stream(eq_unit)), "cosine_topk")
    return (scores, idx, status) if return_status else (scores, idx)


def packed_result_bytes(Q: int, k: int) -> int:
    """Bytes of one rank's result buffer: [Q,k] float32 scores, then (8-byte aligned) [Q,k] int64 indices."""
    return (Q * k * 4 + 7) // 8 * 8 + Q * k * 8


def packed_result_views(buf: torch.Tensor, Q: int, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(scores, idx) views of result buffers ``buf`` uint8 [..., packed_result_bytes(Q, k)] (one per leading index): what
    :func:`cosine_topk` writes through ``out=`` and what :func:`topk_merge` reads in place after an all-gather."""
    so = (Q * k * 4 + 7) // 8 * 8
    s = buf[..., :Q * k * 4].view(torch.float32)
    i = buf[..., so:so + Q * k * 8].view(torch.int64)
    return s.unflatten(-1, (Q, k)), i.unflatten(-1, (Q, k))


def _list_major(t: torch.Tensor) -> bool:
    """[nlists, Q, k] with each list contiguous (lists may be any distance apart)"""
    return t.dim() == 3 and t.stride(2) == 1 and t.stride(1) == t.shape[2] and (t.shape[0] == 1 or t.stride(0) >= t.shape[1] * t.shape[2])


def topk_merge(scores, idx, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge per-shard/per-chunk lists (global indices) into [Q,k].  ``scores`` / ``idx`` are either sequences of [Q,k_in]
    tensors or [nlists, Q, k_in] tensors; the latter are read in place when every list is contiguous, whatever the distance
    between lists (e.g. :func:`packed_result_views` of an all-gathered buffer)."""
    s = scores if isinstance(scores, torch.Tensor) else torch.stack([t.contiguous() for t in scores])
    i = idx if isinstance(idx, torch.Tensor) else torch.stack([t.contiguous() for t in idx])
    if not _list_major(s):
        s = s.contiguous()
    if not _list_major(i):
        i = i.contiguous()
    _need_gpu(s, i)
    if s.dtype != torch.float32 or i.dtype != torch.int64 or s.shape != i.shape or s.dim() != 3:
        raise ValueError("topk_merge expects float32 scores and int64 indices of one shape [nlists, Q, k_in]")
    nl, Q, k_in = s.shape
    out_s = torch.empty((Q, k), dtype=torch.float32, device=s.device)
    out_i = torch.empty((Q, k), dtype=torch.int64, device=s.device)
    with torch.cuda.device(s.device):
        _lib.check(_lib.lib().tsim_topk_merge_strided(s.data_ptr(), i.data_ptr(), nl, Q, k_in, k, s.stride(0), i.stride(0),
                                                      out_s.data_ptr(), out_i.data_ptr(), _stream(s)), "topk_merge")
    return out_s, out_i


def cos_sim_dense(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _need_gpu(a, b)
    a = a.float().contiguous()
    b = b.float().contiguous()
    if a.shape[1] != b.shape[1]:
        raise ValueError("cos_sim: width mismatch")
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(_lib.lib().tsim_cos_sim(a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], a.shape[1],
                                           out.data_ptr(), _stream(a)), "cos_sim")
    return out


def mean_pool(hidden: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    _need_gpu(hidden, mask)
    assert len(hidden.shape) == 3  # batch, seq_len, embed_size (modules.py:159)
    if hidden.dtype not in (torch.float32, torch.bfloat16):
        hidden = hidden.float()
    hidden = hidden.contiguous()
    m = mask.to(torch.int32).contiguous()
    B, S, H = hidden.shape
    out = torch.empty((B, H), dtype=torch.float32, device=hidden.device)
    dt = _lib.TSIM_F32 if hidden.dtype == torch.float32 else _lib.TSIM_BF16
    with torch.cuda.device(hidden.device):
        _lib.check(_lib.lib().tsim_mean_pool(hidden.data_ptr(), dt, m.data_ptr(), B, S, H, out.data_ptr(),
                                             _stream(hidden)), "mean_pool")
    return out


def quantize_mxfp8(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """[rows, K] bf16 -> (e4m3 bytes uint8 [rows, K], E8M0 block scales uint8 [rows, K/32]) — the operand format of
    the fp8 encoder variant (``NativeEncoder(weight_dtype="mxfp8")``).  K % 32 == 0."""
    _need_gpu(x)
    if x.dim() != 2 or x.dtype != torch.bfloat16:
        raise ValueError("quantize_mxfp8 expects a 2-D bfloat16 tensor")
    rows, K = x.shape
    if K % 32 != 0:
        raise ValueError(f"K={K} is not a multiple of the 32-element MX block")
    x = x.contiguous()
    q = torch.empty((rows, K), dtype=torch.uint8, device=x.device)
    s = torch.empty((rows, K // 32), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().tsim_quantize_mxfp8(x.data_ptr(), rows, K, q.data_ptr(), s.data_ptr(), _stream(x)),
                   "quantize_mxfp8")
    return q, s


def gemm_mxfp8(xq: torch.Tensor, xs: torch.Tensor, wq: torch.Tensor, ws: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """float32 [M, N] = dequant(xq, xs) @ dequant(wq, ws)^T + bias with the block-scaled fp8 MFMA.
    xq [M, K] / wq [N, K] uint8 e4m3 bytes, xs [M, K/32] / ws [N, K/32] uint8 E8M0 scales, bias float32 [N]."""
    _need_gpu(xq, xs, wq, ws, bias)
    M, K = xq.shape
    N = wq.shape[0]
    if wq.shape[1] != K or xs.shape != (M, K // 32) or ws.shape != (N, K // 32) or bias.shape != (N,):
        raise ValueError("gemm_mxfp8: inconsistent operand shapes")
    Mp = (M + 255) // 256 * 256                      # the kernel works on whole 256-row tiles
    xqp = torch.zeros((Mp, K), dtype=torch.uint8, device=xq.device)
    xsp = torch.full((Mp, K // 32), 127, dtype=torch.uint8, device=xq.device)
    xqp[:M], xsp[:M] = xq, xs
    out = torch.empty((Mp, N), dtype=torch.float32, device=xq.device)
    with torch.cuda.device(xq.device):
        _lib.check(_lib.lib().tsim_gemm_mxfp8(xqp.data_ptr(), xsp.data_ptr(), wq.contiguous().data_ptr(),
                                              ws.contiguous().data_ptr(), bias.contiguous().data_ptr(), out.data_ptr(),
                                              M, N, K, _stream(xq)), "gemm_mxfp8")
    return out[:M]
