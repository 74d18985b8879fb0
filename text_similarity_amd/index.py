"""GpuFlatIndex — an exact cosine index resident in HBM with the call surface the reference uses from
``hnswlib.Index(space='cosine')`` (/root/reference/src/pipeline/search_pipeline.py:105-169): ``init_index``,
``add_items``, ``knn_query``, ``mark_deleted``, ``resize_index``, ``save_index`` / ``load_index``, ``get_current_count``,
``set_ef``.  Every query is a brute-force pass of the fused MFMA cosine + top-k kernel, so results are exact (hnswlib's are
approximate) and ordered by (score desc, label-row asc).

Layout: float32 rows ``[capacity, d]`` as given (the reference's scores are cosines of these) + their unit float16 rows
``[capacity, pad_dim(d)]`` (what the MFMA kernel streams to select candidates) + int64 labels + a tombstone mask.
Deleting marks a tombstone; the matrices are compacted (one device gather) before the next query, so deleted rows cost
nothing afterwards and labels stay stable.  On disk (``index.bin``, a numpy ``.npz`` written without pickling): the
float32 live rows, their labels, ``d`` (unit rows are recomputed on load; a file written by the first version of this
index holds bf16 unit rows only, which then ARE the float32 rows).  k <= 64, d <= 768.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops


class GpuFlatIndex:
    def __init__(self, space: str = "cosine", dim: int = 0, device: Optional[torch.device] = None):
        if space != "cosine":
            raise ValueError("GpuFlatIndex implements the cosine space only")
        self.dim = int(dim)
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        self._rows: Optional[torch.Tensor] = None      # [capacity, ld] float16 unit rows
        self._f32: Optional[torch.Tensor] = None       # [capacity, d] float32 rows as given
        self._labels: Optional[torch.Tensor] = None    # [capacity] int64
        self._dead: Optional[torch.Tensor] = None      # [capacity] bool
        self._rho: Optional[torch.Tensor] = None       # [1] float32: largest rounding residual of any unit row ever stored
        self._n = 0
        self._n_dead = 0

    # ------------------------------------------------------------------ hnswlib-shaped surface
    def init_index(self, max_elements: int, ef_construction: int = 0, M: int = 0):
        self._reserve(int(max_elements))

    def set_ef(self, ef: int):      # exact search: nothing to tune
        pass

    def resize_index(self, new_size: int):
        self._reserve(int(new_size))

    def get_current_count(self) -> int:
        """rows ever added and not yet compacted away + live rows == hnswlib's count of inserted elements"""
        return self._n

    def num_live(self) -> int:
        return self._n - self._n_dead

    def add_items(self, data, ids: Optional[Iterable[int]] = None, num_threads: int = -1):
        x = torch.as_tensor(np.asarray(data) if not isinstance(data, torch.Tensor) else data)
        if x.dim() == 1:
            x = x.unsqueeze(0)
        if self.dim == 0:
            self.dim = int(x.shape[1])
        if x.shape[1] != self.dim:
            raise ValueError(f"expected width {self.dim}, got {x.shape[1]}")
        n = x.shape[0]
        if ids is None:
            ids = np.arange(self._n, self._n + n)
        lab = torch.as_tensor(np.asarray(list(ids), dtype=np.int64))
        if lab.numel() != n:
            raise ValueError("ids and data disagree in length")
        xf = x.to(self.device, dtype=torch.float32).contiguous()
        if self._rho is None:
            self._rho = ops.new_rho(self.device)
        unit = ops.l2norm_rows(xf, rho=self._rho)       # the word only grows: deleted rows leave the bound conservative
        self._reserve(self._n + n)
        self._rows[self._n:self._n + n] = unit
        self._f32[self._n:self._n + n] = xf
        self._labels[self._n:self._n + n] = lab.to(self.device)
        self._dead[self._n:self._n + n] = False
        self._n += n

    def mark_deleted(self, label: int):
        if self._n == 0:
            raise RuntimeError("label not found")
        hit = (self._labels[:self._n] == int(label)) & ~self._dead[:self._n]
        k = int(hit.sum())
        if k == 0:
            raise RuntimeError("label not found")      # hnswlib raises RuntimeError too (search_pipeline.py:167)
        self._dead[:self._n] |= hit
        self._n_dead += k

    def knn_query(self, data, k: int = 1) -> Tuple[np.ndarray, np.ndarray]:
        """(labels [Q,k] int64, distances [Q,k] float32 = 1 - cosine), best first — hnswlib's return convention."""
        labels, scores = self.search(data, k)
        return labels.cpu().numpy(), (1.0 - scores).cpu().numpy()

    # ------------------------------------------------------------------ device-level API
    def search(self, data, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(labels [Q,k] int64, scores [Q,k] float32) on the device; -1 / -inf pad when fewer than k live rows."""
        self._compact()
        q = torch.as_tensor(np.asarray(data) if not isinstance(data, torch.Tensor) else data)
        if q.dim() == 1:
            q = q.unsqueeze(0)
        qf = q.to(self.device, dtype=torch.float32).contiguous()
        qn = ops.l2norm_rows(qf)
        if self._n == 0:
            Q = q.shape[0]
            return (torch.full((Q, k), -1, dtype=torch.int64, device=self.device),
                    torch.full((Q, k), float("-inf"), device=self.device))
        s, i = ops.cosine_topk(qn, self._rows[:self._n], self.dim, k, eq_f32=qf, ec_f32=self._f32[:self._n], rho_c=self._rho)
        lab = torch.where(i >= 0, self._labels[i.clamp(min=0)], torch.full_like(i, -1))
        return lab, s

    # ------------------------------------------------------------------ persistence
    def save_index(self, path: str):
        self._compact()
        if os.path.isdir(path):
            path = os.path.join(path, "index.bin")
        labels = self._labels[:self._n].cpu().numpy() if self._n else np.zeros((0,), np.int64)
        rows = self._f32[:self._n].cpu().numpy() if self._n else np.zeros((0, self.dim), np.float32)
        with open(path, "wb") as f:
            np.savez(f, rows_f32=rows, labels=labels, dim=np.int64(self.dim))

    def load_index(self, path: str, max_elements: int = 0):
        if os.path.isdir(path):
            path = os.path.join(path, "index.bin")
        z = np.load(path, allow_pickle=False)
        self.dim = int(z["dim"])
        labels = z["labels"]
        self._rows = self._f32 = self._labels = self._dead = None
        self._rho = ops.new_rho(self.device)
        self._n = self._n_dead = 0
        if "rows_f32" in z.files:
            rows = z["rows_f32"]
        else:   # first-version file: bf16 bit patterns of unit rows
            rows = (z["rows_bf16"].astype(np.uint32) << np.uint32(16)).view(np.float32)
        n = rows.shape[0]
        self._reserve(max(n, int(max_elements)))
        if n:
            xf = torch.from_numpy(np.ascontiguousarray(rows)).to(self.device)
            self._f32[:n] = xf
            self._rows[:n] = ops.l2norm_rows(xf, rho=self._rho)
        if n:
            self._labels[:n] = torch.from_numpy(labels).to(self.device)
            self._n = n

    # ------------------------------------------------------------------ internals
    def _reserve(self, n: int):
        if self.dim == 0:
            return
        cap = 0 if self._rows is None else self._rows.shape[0]
        if n <= cap:
            return
        new_cap = max(n, 2 * cap, 1024)
        ld = ops.pad_dim(self.dim)
        rows = torch.zeros((new_cap, ld), dtype=ops.UNIT_DTYPE, device=self.device)
        f32 = torch.zeros((new_cap, self.dim), dtype=torch.float32, device=self.device)
        if self._f32 is not None and self._n:
            f32[:self._n] = self._f32[:self._n]
        self._f32 = f32
        labels = torch.full((new_cap,), -1, dtype=torch.int64, device=self.device)
        dead = torch.zeros((new_cap,), dtype=torch.bool, device=self.device)
        if self._rows is not None and self._n:
            rows[:self._n] = self._rows[:self._n]
            labels[:self._n] = self._labels[:self._n]
            dead[:self._n] = self._dead[:self._n]
        self._rows, self._labels, self._dead = rows, labels, dead

    def _compact(self):
        if self._n_dead == 0:
            return
        keep = (~self._dead[:self._n]).nonzero(as_tuple=False).squeeze(1)
        m = keep.numel()
        self._rows[:m] = self._rows[keep]
        self._f32[:m] = self._f32[keep]
        self._labels[:m] = self._labels[keep]
        self._dead[:self._n] = False
        self._n, self._n_dead = m, 0
