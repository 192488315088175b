"""CPU oracle for the ALPINE multiplicative-update NMF fit loop.

TEST INFRASTRUCTURE ONLY.  Nothing under ``alpine_amd/`` imports this module; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may.
It is the checker the HIP path is compared against, never the thing shipped or measured
as the product.

It restates, in torch-CPU float32 (the arithmetic the reference itself uses on
``device="cpu"``), the hot path of ylaboratory/ALPINE v0.2.0:

* ``init_factors``      <- ``ALPINE._initialize_matrices``   alpine/main.py:436-472
* ``orth_matrix``       <- ``ALPINE._compute_orthogonal_matrix`` alpine/main.py:474-484
* ``mu_step_faithful``  <- MU branch of ``ALPINE._fit``       alpine/main.py:589-663
                           (incl. the per-epoch randperm gather, main.py:502-521,
                           alpine/utils/sampling.py:6-16, :58-71)
* ``als_step_faithful`` <- block-coordinate branch (use_als)   alpine/main.py:523-588
* ``loss_row``          <- ``ALPINE._compute_loss``           alpine/main.py:726-753
* ``scale_factors``     <- ``ALPINE._scale_matrices``         alpine/main.py:772-781
* ``transform_faithful``<- loop of ``ALPINE._transform``      alpine/main.py:705-709
* ``fit_faithful``      <- the loop ``ALPINE._fit``           alpine/main.py:486-676
* ``mu_step_fused`` / ``fit_fused``: the SAME mathematics re-associated the way the HIP
  kernels evaluate it (``W(HH^T)``, ``(W^TW)H``, identity cell order, trace-form loss);
  this is the numerical specification the kernels are written to.

Pinning: the reference ships no tests and no golden vectors, so this
restatement is pinned against outputs of the reference itself, run in the build container
by ``oracle/gen_golden.py`` and committed under ``tests/golden/`` (see that script and
``tests/test_oracle_golden.py``).

State layout used here (differs from the reference's lists of per-group tensors):
``W`` is one G x K tensor, ``H`` one K x N tensor, with the K columns/rows ordered
``[cov_1 | cov_2 | ... | unguided]`` exactly as ``n_all_components`` orders them
(main.py:79); ``Bs[i]`` is C_i x k_i and ``Ys[i]`` is C_i x N (main.py:446-449).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch


@dataclass
class OracleParams:
    n_components: int
    n_covariate_components: List[int]
    lam: List[float]
    orth_W: float = 0.0
    alpha_W: float = 0.0
    l1_ratio_W: float = 0.0
    loss_type: str = "kl-divergence"
    eps: float = 1e-6
    random_state: int = 42
    use_als: bool = False

    @property
    def n_all_components(self) -> List[int]:
        return list(self.n_covariate_components) + [self.n_components]

    @property
    def total_components(self) -> int:
        return sum(self.n_all_components)

    @property
    def offsets(self) -> List[int]:
        off, out = 0, []
        for k in self.n_all_components:
            out.append(off)
            off += k
        return out


@dataclass
class OracleState:
    X: torch.Tensor            # G x N
    Ys: List[torch.Tensor]     # C_i x N
    W: torch.Tensor            # G x K
    H: torch.Tensor            # K x N
    Bs: List[torch.Tensor]     # C_i x k_i
    losses: List[List[float]] = field(default_factory=list)

    def clone(self) -> "OracleState":
        return OracleState(self.X, self.Ys, self.W.clone(), self.H.clone(),
                           [b.clone() for b in self.Bs], [list(r) for r in self.losses])


# --------------------------------------------------------------------------- init
def init_factors(p: OracleParams, X_gn: np.ndarray, Ys_nc: Sequence[np.ndarray]) -> OracleState:
    """main.py:436-472.  Draw order matters: every W_j, then every H_j, then every B_i, each
    a separate ``torch.rand`` call on the default CPU generator reseeded with random_state,
    each clamped from below at eps."""
    torch.manual_seed(p.random_state)
    X = torch.tensor(np.ascontiguousarray(X_gn), dtype=torch.float32)
    Ys = [torch.tensor(np.ascontiguousarray(np.asarray(y).T), dtype=torch.float32) for y in Ys_nc]
    G, N = X.shape
    Ws = [torch.rand((G, k), dtype=torch.float32).clamp(min=p.eps) for k in p.n_all_components]
    Hs = [torch.rand((k, N), dtype=torch.float32).clamp(min=p.eps) for k in p.n_all_components]
    Bs = [torch.rand((y.shape[0], k), dtype=torch.float32).clamp(min=p.eps)
          for y, k in zip(Ys, p.n_covariate_components)]
    return OracleState(X, Ys, torch.cat(Ws, dim=1), torch.cat(Hs, dim=0), Bs)


def orth_matrix(p: OracleParams, K: int) -> torch.Tensor:
    """main.py:474-484: orth_W * (ones(K,K) - eye(K))."""
    return p.orth_W * (torch.ones((K, K), dtype=torch.float32) - torch.eye(K, dtype=torch.float32))


# ------------------------------------------------------------------ faithful step
def mu_step_faithful(p: OracleParams, s: OracleState, perm: Optional[torch.Tensor]) -> None:
    """One full-batch MU iteration with the reference's association and temporaries
    (main.py:589-663).  ``perm`` is the epoch permutation (sampling.py:14); ``None`` means
    identity order.  Python precedence is kept: ``2 * A @ B`` is ``(2A) @ B`` and
    ``lam * B.T @ Z`` is ``(lam B.T) @ Z``."""
    eps = p.eps
    offs, ks = p.offsets, p.n_all_components
    n_cov = len(p.n_covariate_components)
    if perm is None:
        Xb, Yb, Hb = s.X, list(s.Ys), s.H.clone()
    else:
        Xb = s.X[:, perm]                                   # main.py:520
        Yb = [y[:, perm] for y in s.Ys]                     # main.py:521
        Hb = s.H[:, perm]                                   # main.py:593-594 (cat of gathers)
    W = s.W                                                 # main.py:592 (cat == our storage)
    Hi_old = [Hb[offs[i]:offs[i] + ks[i]].clone() for i in range(n_cov)]   # Hs_batch[i], pre-update

    # --- W update, main.py:596-605
    num = 2 * Xb @ Hb.T
    den = 2 * W @ Hb @ Hb.T + (1 - p.l1_ratio_W) * p.alpha_W * W + W @ orth_matrix(p, W.shape[1])
    den += p.l1_ratio_W * p.alpha_W * torch.ones_like(den)
    den = torch.clamp(den, min=eps)
    W = W * (num / den)
    s.W = W

    # --- B updates, main.py:615-628 (old H, old B)
    for i in range(n_cov):
        Y, Hh, B = Yb[i], Hi_old[i], s.Bs[i]
        if p.loss_type == "kl-divergence":
            num = p.lam[i] * (Y / torch.clamp(B @ Hh, min=eps)) @ Hh.T
            den = p.lam[i] * torch.ones_like(Y) @ Hh.T
        else:
            num = 2 * Y @ Hh.T
            den = 2 * B @ Hh @ Hh.T
        den = torch.clamp(den, min=eps)
        s.Bs[i] = B * (num / den)

    # --- H update, main.py:631-656 (new W, new B, old H)
    num = torch.zeros_like(Hb)
    den = torch.zeros_like(Hb)
    for i in range(n_cov):
        a, b = offs[i], offs[i] + ks[i]
        B, Y, Hh = s.Bs[i], Yb[i], Hi_old[i]
        if p.loss_type == "kl-divergence":
            num[a:b] = p.lam[i] * B.T @ (Y / torch.clamp(B @ Hh, min=eps))
            den[a:b] = p.lam[i] * B.T @ torch.ones_like(Y)
        else:
            num[a:b] = 2 * p.lam[i] * B.T @ Y
            den[a:b] = 2 * p.lam[i] * B.T @ (B @ Hh)
    num += 2 * W.T @ Xb
    den += 2 * W.T @ (W @ Hb)
    den = torch.clamp(den, min=eps)
    Hb = Hb * (num / den)
    if perm is None:
        s.H = Hb
    else:
        s.H[:, perm] = Hb                                   # main.py:659-663


def als_step_faithful(p: OracleParams, s: OracleState, perm: Optional[torch.Tensor]) -> None:
    """The block-coordinate branch, main.py:523-588 (``use_als=True``): for every component group j in the order
    [cov_1 .. cov_C, unguided]: W_j (orthogonality mask of size k_j, main.py:537), then B_j (covariate groups), then H_j,
    each against the CURRENT concatenated W and H (groups < j already updated in this pass)."""
    eps = p.eps
    offs, ks = p.offsets, p.n_all_components
    n_cov = len(p.n_covariate_components)
    if perm is None:
        perm = torch.arange(s.X.shape[1])
    Xb = s.X[:, perm]
    Yb = [y[:, perm] for y in s.Ys]
    for j in range(len(ks)):
        a, b = offs[j], offs[j] + ks[j]
        Hcat = s.H[:, perm]                                   # main.py:527-531 (re-gathered for every group)
        Hj = Hcat[a:b]
        W = s.W[:, a:b]
        num = 2 * Xb @ Hj.T
        den = (2 * s.W @ Hcat @ Hj.T + (1 - p.l1_ratio_W) * p.alpha_W * W @ torch.eye(ks[j], dtype=torch.float32)
               + W @ orth_matrix(p, ks[j]))
        den += p.l1_ratio_W * p.alpha_W * torch.ones_like(den)
        den = torch.clamp(den, min=eps)
        s.W = s.W.clone()
        s.W[:, a:b] = W * (num / den)
        if j < n_cov:                                          # main.py:548-562
            Y, B = Yb[j], s.Bs[j]
            if p.loss_type == "kl-divergence":
                num = p.lam[j] * (Y / torch.clamp(B @ Hj, min=eps)) @ Hj.T
                den = p.lam[j] * torch.ones_like(Y) @ Hj.T
            else:
                num = 2 * Y @ Hj.T
                den = 2 * B @ Hj @ Hj.T
            s.Bs[j] = B * (num / torch.clamp(den, min=eps))
        Wj = s.W[:, a:b]                                       # main.py:565-588
        unum = 2 * Wj.T @ Xb
        uden = 2 * Wj.T @ (s.W @ Hcat)
        if j < n_cov:
            Y, B = Yb[j], s.Bs[j]
            if p.loss_type == "kl-divergence":
                gnum = p.lam[j] * B.T @ (Y / torch.clamp(B @ Hj, min=eps))
                gden = p.lam[j] * B.T @ torch.ones_like(Y)
            else:
                gnum = 2 * p.lam[j] * B.T @ Y
                gden = 2 * p.lam[j] * B.T @ (B @ Hj)
            newH = Hj * ((unum + gnum) / torch.clamp(uden + gden, min=eps))
        else:
            newH = Hj * (unum / torch.clamp(uden, min=eps))
        Hfull = s.H[a:b].clone()
        Hfull[:, perm] = newH
        s.H = s.H.clone()
        s.H[a:b] = Hfull


def loss_row(p: OracleParams, s: OracleState) -> List[float]:
    """main.py:726-753: [total, recon, pred_1..pred_C] as Python floats."""
    eps = p.eps
    recon = (torch.norm(s.X - s.W @ s.H, p="fro") ** 2).item()
    preds = []
    offs, ks = p.offsets, p.n_all_components
    for i in range(len(s.Ys)):
        Hh = s.H[offs[i]:offs[i] + ks[i]]
        if p.loss_type == "kl-divergence":
            y = s.Ys[i]
            y_hat = torch.clamp(s.Bs[i] @ Hh, min=eps)
            preds.append(torch.sum(y * torch.log(torch.clamp(y / y_hat, min=eps)) - y + y_hat).item())
        else:
            preds.append((torch.norm(s.Ys[i] - s.Bs[i] @ Hh, p="fro") ** 2).item())
    total = recon + sum(p.lam[i] * pl for i, pl in enumerate(preds))
    return [total, recon] + preds


def fit_faithful(p: OracleParams, s: OracleState, max_iter: int, use_perm: bool = True,
                 with_loss: bool = True) -> OracleState:
    """main.py:500-667 with batch_size=None, sampling_method="random".  With
    ``use_perm=True`` it draws ``torch.randperm(N)`` once per iteration from the global
    generator exactly where the reference does (sampling.py:14), so that, started right
    after ``init_factors``, it reproduces the reference's index stream."""
    with torch.no_grad():
        N = s.X.shape[1]
        for _ in range(max_iter):
            perm = torch.randperm(N) if use_perm else None
            (als_step_faithful if p.use_als else mu_step_faithful)(p, s, perm)
            if with_loss:
                s.losses.append(loss_row(p, s))
    return s


def balanced_joint_weights(Ys_cn: Sequence[torch.Tensor]) -> np.ndarray:
    """alpine/utils/sampling.py:36-55 (joint label = argmax level of every covariate, NaN rows -> level 0) and
    sklearn's compute_sample_weight("balanced") as used at sampling.py:23: n / (n_classes * count(label))."""
    codes = np.stack([torch.argmax(y, dim=0).numpy() for y in Ys_cn], axis=1)
    _, inv, cnt = np.unique(codes, axis=0, return_inverse=True, return_counts=True)
    return (codes.shape[0] / (len(cnt) * cnt.astype(np.float64)))[np.asarray(inv).reshape(-1)]


def fit_faithful_batches(p: OracleParams, s: OracleState, max_iter: int, batch_size: Optional[int],
                         sampling_method: str = "random", with_loss: bool = True) -> OracleState:
    """main.py:500-667 in full generality: per epoch one index stream (sampling.py:6-33: randperm, or the weighted
    sampler with replacement = torch.multinomial on float64 weights), cut into batches (sampling.py:58-71); every
    batch runs the MU step on the gathered columns and scatters H back; one loss row per epoch over all cells."""
    N = s.X.shape[1]
    bs = N if batch_size is None else batch_size
    w = torch.as_tensor(balanced_joint_weights(s.Ys), dtype=torch.double) if sampling_method == "weighted" else None
    with torch.no_grad():
        for _ in range(max_iter):
            epoch = torch.multinomial(w, N, True) if w is not None else torch.randperm(N)
            for b0 in range(0, N, bs):
                (als_step_faithful if p.use_als else mu_step_faithful)(p, s, epoch[b0:min(b0 + bs, N)])
            if with_loss:
                s.losses.append(loss_row(p, s))
    return s


# --------------------------------------------------------------------- fused step
def fused_reduce_terms(p: OracleParams, s: OracleState):
    """Everything one iteration needs that is a SUM OVER CELLS of per-cell quantities of the
    OLD H (and old B): these are the terms the sharded build all-reduces."""
    eps = p.eps
    offs, ks = p.offsets, p.n_all_components
    XHt = s.X @ s.H.T                       # G x K
    HHt = s.H @ s.H.T                       # K x K
    bnum, bden, pred = [], [], []
    for i in range(len(s.Ys)):
        Hh, Y, B = s.H[offs[i]:offs[i] + ks[i]], s.Ys[i], s.Bs[i]
        if p.loss_type == "kl-divergence":
            yhat = torch.clamp(B @ Hh, min=eps)
            bnum.append((p.lam[i] * (Y / yhat)) @ Hh.T)                 # C x k
            bden.append((p.lam[i] * Hh).sum(dim=1))                      # k   (lam*1 @ Hh.T rows are equal)
            pred.append(torch.sum(Y * torch.log(torch.clamp(Y / yhat, min=eps)) - Y + yhat).double())
        else:
            bnum.append(Y @ Hh.T)
            bden.append(torch.zeros(ks[i]))
            pred.append(torch.sum((Y - B @ Hh) ** 2).double())
    return XHt, HHt, bnum, bden, pred


def mu_step_fused(p: OracleParams, s: OracleState, terms=None) -> None:
    """One MU iteration in the association the HIP kernels use:
    ``den_W = W @ (2 HH^T + orth + (1-rho) alpha I) + rho alpha``,
    ``den_H = (2 W^T W) @ H + guided``; identity cell order (a full-batch permutation only
    re-orders the sums)."""
    eps = p.eps
    offs, ks = p.offsets, p.n_all_components
    n_cov = len(p.n_covariate_components)
    K = s.W.shape[1]
    XHt, HHt, bnum, bden, _ = terms if terms is not None else fused_reduce_terms(p, s)

    M = 2 * HHt + orth_matrix(p, K) + (1 - p.l1_ratio_W) * p.alpha_W * torch.eye(K)
    den = torch.clamp(s.W @ M + p.l1_ratio_W * p.alpha_W, min=eps)
    s.W = s.W * ((2 * XHt) / den)

    for i in range(n_cov):
        B = s.Bs[i]
        if p.loss_type == "kl-divergence":
            num = bnum[i]
            den = bden[i].unsqueeze(0).expand_as(B)
        else:
            a, b = offs[i], offs[i] + ks[i]
            num = 2 * bnum[i]
            den = (2 * B) @ HHt[a:b, a:b]
        s.Bs[i] = B * (num / torch.clamp(den, min=eps))

    WtW2 = 2 * (s.W.T @ s.W)
    num = 2 * (s.W.T @ s.X)
    den = WtW2 @ s.H
    for i in range(n_cov):
        a, b = offs[i], offs[i] + ks[i]
        B, Y, Hh = s.Bs[i], s.Ys[i], s.H[a:b]
        if p.loss_type == "kl-divergence":
            lamBt = p.lam[i] * B.T
            num[a:b] += lamBt @ (Y / torch.clamp(B @ Hh, min=eps))
            den[a:b] += lamBt.sum(dim=1, keepdim=True)
        else:
            lamBt = 2 * p.lam[i] * B.T
            num[a:b] += lamBt @ Y
            den[a:b] += lamBt @ (B @ Hh)
    s.H = s.H * (num / torch.clamp(den, min=eps))


def trace_loss_row(p: OracleParams, s: OracleState, xnorm2: float, terms=None) -> List[float]:
    """Loss row of the CURRENT factors from the reduce terms of the current H (trace form,
    float64 finalise): ``||X||^2 - 2<XH^T, W> + <W^TW, HH^T>``."""
    XHt, HHt, _, _, pred = terms if terms is not None else fused_reduce_terms(p, s)
    WtW = (s.W.T @ s.W).double()
    recon = xnorm2 - 2.0 * torch.sum(XHt.double() * s.W.double()).item() + torch.sum(WtW * HHt.double()).item()
    preds = [float(x) for x in pred]
    total = recon + sum(p.lam[i] * pl for i, pl in enumerate(preds))
    return [total, recon] + preds


def fit_fused(p: OracleParams, s: OracleState, max_iter: int, with_loss: bool = True) -> OracleState:
    with torch.no_grad():
        xnorm2 = torch.sum(s.X.double() ** 2).item()
        terms = fused_reduce_terms(p, s)
        for _ in range(max_iter):
            mu_step_fused(p, s, terms)
            terms = fused_reduce_terms(p, s)          # terms of the new H: next step + this step's loss
            if with_loss:
                s.losses.append(trace_loss_row(p, s, xnorm2, terms))
    return s


# ------------------------------------------------------------------------- scaling
def scale_factors(p: OracleParams, s: OracleState) -> None:
    """main.py:772-781: every column of W is divided by its sum, the matching row of H is
    multiplied by it and the matching column of B_i divided by it."""
    offs, ks = p.offsets, p.n_all_components
    Wn, Hn = s.W.clone(), s.H.clone()
    for j, (a, k) in enumerate(zip(offs, ks)):
        sc = s.W[:, a:a + k].sum(dim=0)
        Wn[:, a:a + k] = s.W[:, a:a + k] / sc
        Hn[a:a + k] = s.H[a:a + k] * sc.unsqueeze(1)
        if j < len(p.n_covariate_components):
            s.Bs[j] = s.Bs[j] / sc
    s.W, s.H = Wn, Hn


# ----------------------------------------------------------------------- transform
def transform_faithful(eps: float, W: torch.Tensor, X_gn: torch.Tensor, H0: torch.Tensor, n_iter: int) -> torch.Tensor:
    """main.py:705-709: ``H *= 2 W^T X / clamp(2 W^T (W H), eps)`` n_iter times with W frozen (H0 is the unclamped,
    unseeded ``torch.rand`` draw of main.py:687-689; the caller provides it so the RNG stream can be mirrored)."""
    H = H0.clone()
    with torch.no_grad():
        for _ in range(n_iter):
            num = 2 * W.T @ X_gn
            den = 2 * W.T @ (W @ H)
            den = torch.clamp(den, min=eps)
            H *= num / den
    return H


# ---------------------------------------------------------------- common evaluator
def recon_loss_f64(X_gn: np.ndarray, W: np.ndarray, H: np.ndarray, block: int = 4096) -> float:
    """Accurate ||X - WH||_F^2 (float64 accumulation, blocked over cells).  The 'common
    evaluator' of SURVEY.md section 7: torch-CPU fp32 ``torch.norm(R)**2`` is biased low by
    percents at the BASELINE shapes, so factor sets are scored with this instead."""
    W64 = np.asarray(W, dtype=np.float64)
    acc = 0.0
    for n0 in range(0, X_gn.shape[1], block):
        R = np.asarray(X_gn[:, n0:n0 + block], dtype=np.float64) - W64 @ np.asarray(H[:, n0:n0 + block], dtype=np.float64)
        acc += float(np.sum(R * R))
    return acc


def split_groups(p: OracleParams, W: torch.Tensor, H: torch.Tensor):
    """Back to the reference's per-group lists (``Ws``/``Hs`` of ``AlpineMatrices``, main.py:28-34)."""
    Ws, Hs = [], []
    for a, k in zip(p.offsets, p.n_all_components):
        Ws.append(W[:, a:a + k])
        Hs.append(H[a:a + k])
    return Ws, Hs
