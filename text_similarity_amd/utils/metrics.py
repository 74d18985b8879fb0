"""``cos_sim`` (/root/reference/src/utils/metrics.py:81-101): dense cosine matrix, computed by the HIP kernel
behind ``tsim_cos_sim``.  Same promotion rules: non-tensors are converted, 1-D inputs become one row; no eps, so a
zero row yields NaN exactly like the reference."""
from __future__ import annotations

import torch

from .. import ops


def cos_sim(a, b, device=None):
    if not isinstance(a, torch.Tensor):
        a = torch.tensor(a)
    if not isinstance(b, torch.Tensor):
        b = torch.tensor(b)
    if len(a.shape) == 1:
        a = a.unsqueeze(0)
    if len(b.shape) == 1:
        b = b.unsqueeze(0)
    dev = device or (a.device if a.is_cuda else (b.device if b.is_cuda else torch.device("cuda")))
    return ops.cos_sim_dense(a.to(dev), b.to(dev))


class AverageMeter:
    """Running average with the reference's fields (/root/reference/src/utils/metrics.py AverageMeter: val, sum,
    count, avg) — the base of the meters below."""

    def __init__(self, name: str = "metric", **kwargs):
        self.name = name
        self.reset()

    def reset(self):
        self.val = self.sum = self.count = self.avg = 0

    def update(self, val, n=1, **kwargs):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def top1_matches(src_embeddings: torch.Tensor, tgt_embeddings: torch.Tensor):
    """For every source row the index (and score) of its most similar target row — the ``np.argmax(cos_sims[i])`` of
    /root/reference/src/utils/metrics.py:476-477 without materialising the N x N matrix: one fused cosine top-1 call.
    Ties go to the lower index, as numpy's argmax does."""
    d = src_embeddings.shape[1]
    dev = src_embeddings.device if src_embeddings.is_cuda else (tgt_embeddings.device if tgt_embeddings.is_cuda else torch.device("cuda"))
    sf = src_embeddings.to(dev).float().contiguous()
    tf = tgt_embeddings.to(dev).float().contiguous()
    # scores and order are the float32 cosines the reference's cos_sim matrix holds (exact re-score of the MFMA candidates)
    scores, idx = ops.cosine_topk(ops.l2norm_rows(sf), ops.l2norm_rows(tf), d, 1, eq_f32=sf, ec_f32=tf)
    return idx[:, 0], scores[:, 0]


class RetrievalAccuracyMeter(AverageMeter):
    """Bidirectional top-1 retrieval accuracy (/root/reference/src/utils/metrics.py:450-507): fraction of source rows
    whose nearest target row is their own index, and the reverse direction; ``avg`` is their mean.  Two fused top-1
    searches on the GPU replace the dense ``cos_sim`` matrix + per-row ``np.argmax`` loops."""

    def __init__(self, print_wrong_matches: bool = True, **kwargs):
        super().__init__(name="accuracy", **kwargs)
        self.print_wrong_matches = print_wrong_matches
        self.src2tgt = 0
        self.tgt2src = 0
        self.lines = []
        self.precision = self.recall = self.f1 = 0

    def __str__(self):
        accuracy = "accuracy [src2tgt: {:.2f} tgt2src: {:.2f}]".format(self.src2tgt, self.tgt2src)
        f1 = "precision: {:.2f} recall: {:.2f} f1: {:.2f}".format(self.precision, self.recall, self.f1)
        return "\n\n".join(self.lines + [accuracy, f1])

    def update(self, src_embeddings, tgt_embeddings, source_sentences=None, target_sentences=None, **kwargs):
        src = src_embeddings if isinstance(src_embeddings, torch.Tensor) else torch.as_tensor(src_embeddings)
        tgt = tgt_embeddings if isinstance(tgt_embeddings, torch.Tensor) else torch.as_tensor(tgt_embeddings)
        n = src.shape[0]
        own = torch.arange(n, device="cuda")
        fwd_idx, fwd_score = top1_matches(src, tgt)
        bwd_idx, _ = top1_matches(tgt, src)
        own = own.to(fwd_idx.device)
        if self.print_wrong_matches and source_sentences is not None and target_sentences is not None:
            wrong = torch.nonzero(fwd_idx != own).flatten().tolist()
            for i in wrong:
                j = int(fwd_idx[i])
                self.lines.append(f"i: {i} j: {j}, INCORRECT\nsrc: {source_sentences[i]}\ntgt: {target_sentences[j]}\n"
                                  f"maximum score: {float(fwd_score[i])}")
        self.src2tgt = float((fwd_idx == own).sum()) / n
        self.tgt2src = float((bwd_idx == own).sum()) / n
        self.avg = (self.src2tgt + self.tgt2src) / 2
        self.val = self.avg


class EmbeddingSimilarityMeter(AverageMeter):
    """STS correlation meter (/root/reference/src/utils/metrics.py:317-381): Pearson / Spearman correlation of gold
    scores with paired cosine, negative Manhattan, negative Euclidean and dot-product similarities of two embedding
    sets; ``val`` is the best Spearman, as in the reference.  The paired similarities are row reductions on the
    device (torch); the correlations are scipy's, on the host, as in the reference."""

    def __init__(self, **kwargs):
        super().__init__(name="embed_sim", **kwargs)
        for k in ("cosine", "manhattan", "euclidean", "dot"):
            setattr(self, f"eval_pearson_{k}", 0)
            setattr(self, f"eval_spearman_{k}", 0)

    def update(self, embeddings, labels, n, **kwargs):
        from scipy.stats import pearsonr, spearmanr
        e1 = torch.as_tensor(embeddings[0]).float()
        e2 = torch.as_tensor(embeddings[1]).float()
        if torch.cuda.is_available():
            e1, e2 = e1.cuda(), e2.cuda()
        sims = {
            "cosine": torch.nn.functional.cosine_similarity(e1, e2, dim=-1),
            "manhattan": -(e1 - e2).abs().sum(-1),
            "euclidean": -(e1 - e2).pow(2).sum(-1).sqrt(),
            "dot": (e1 * e2).sum(-1),
        }
        labels = [float(x) for x in labels]
        best = -2.0
        for k, v in sims.items():
            v = v.cpu().numpy()
            setattr(self, f"eval_pearson_{k}", pearsonr(labels, v)[0])
            sp = spearmanr(labels, v)[0]
            setattr(self, f"eval_spearman_{k}", sp)
            best = max(best, sp)
        self.val = best
        self.sum += self.val * n
        self.count += n
        self.avg = self.sum / self.count
