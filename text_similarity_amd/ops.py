"""Device ops: thin wrappers that pass raw device pointers of torch tensors to libtsim.so.

torch is used for allocation, stream identity and host<->device copies only.  Every function requires
CUDA (ROCm) tensors and raises otherwise: there is no CPU path in the product."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch

from . import _lib


def _need_gpu(*ts):
    for t in ts:
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise _lib.TsimError("text_similarity_amd ops run on MI355X only: expected a CUDA/ROCm tensor, "
                                 f"got {type(t).__name__} on {getattr(t, 'device', None)}")


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


_workspaces = {}


def _workspace(dev: torch.device, nbytes: int) -> torch.Tensor:
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
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


def l2norm_rows(x: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """[rows, d] float32/bf16 -> unit rows in bf16, zero-padded to [rows, pad_dim(d)] (A7 operand prep)."""
    _need_gpu(x)
    if x.dim() != 2:
        raise ValueError("l2norm_rows expects a 2-D tensor")
    if x.dtype not in (torch.float32, torch.bfloat16):
        x = x.float()
    x = x.contiguous()
    rows, d = x.shape
    ld = pad_dim(d)
    out = torch.empty((rows, ld), dtype=torch.bfloat16, device=x.device)
    dt = _lib.TSIM_F32 if x.dtype == torch.float32 else _lib.TSIM_BF16
    _lib.check(_lib.lib().tsim_l2norm_rows(x.data_ptr(), dt, rows, d, x.stride(0), out.data_ptr(), ld, eps,
                                           _stream(x)), "l2norm_rows")
    return out


def cosine_topk(eq_unit: torch.Tensor, ec_unit: torch.Tensor, d: int, k: int, idx_offset: int = 0
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Top-k inner products of unit bf16 rows (outputs of l2norm_rows): scores [Q,k] f32, idx [Q,k] i64,
    ordered by (score desc, index asc)."""
    _need_gpu(eq_unit, ec_unit)
    if eq_unit.dtype != torch.bfloat16 or ec_unit.dtype != torch.bfloat16:
        raise ValueError("cosine_topk expects bf16 unit rows from l2norm_rows")
    ld = pad_dim(d)
    if eq_unit.shape[1] != ld or ec_unit.shape[1] != ld or not eq_unit.is_contiguous() or not ec_unit.is_contiguous():
        raise ValueError(f"cosine_topk: rows must be contiguous with stride pad_dim({d})={ld}")
    Q, N = eq_unit.shape[0], ec_unit.shape[0]
    scores = torch.empty((Q, k), dtype=torch.float32, device=eq_unit.device)
    idx = torch.empty((Q, k), dtype=torch.int64, device=eq_unit.device)
    if Q == 0:
        return scores, idx
    L = _lib.lib()
    nbytes = L.tsim_cosine_topk_workspace_bytes(Q, N, k)
    ws = _workspace(eq_unit.device, nbytes)
    _lib.check(L.tsim_cosine_topk(eq_unit.data_ptr(), Q, ec_unit.data_ptr(), N, d, ld, k, scores.data_ptr(),
                                  idx.data_ptr(), idx_offset, ws.data_ptr(), ws.numel(), _stream(eq_unit)),
               "cosine_topk")
    return scores, idx


def topk_merge(scores: Sequence[torch.Tensor], idx: Sequence[torch.Tensor], k: int
               ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge per-shard/per-chunk [Q,k_in] lists (global indices) into [Q,k]."""
    s = torch.stack([t.contiguous() for t in scores]).contiguous()
    i = torch.stack([t.contiguous() for t in idx]).contiguous()
    _need_gpu(s, i)
    nl, Q, k_in = s.shape
    out_s = torch.empty((Q, k), dtype=torch.float32, device=s.device)
    out_i = torch.empty((Q, k), dtype=torch.int64, device=s.device)
    _lib.check(_lib.lib().tsim_topk_merge(s.data_ptr(), i.data_ptr(), nl, Q, k_in, k, out_s.data_ptr(),
                                          out_i.data_ptr(), _stream(s)), "topk_merge")
    return out_s, out_i


def cos_sim_dense(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _need_gpu(a, b)
    a = a.float().contiguous()
    b = b.float().contiguous()
    if a.shape[1] != b.shape[1]:
        raise ValueError("cos_sim: width mismatch")
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
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
