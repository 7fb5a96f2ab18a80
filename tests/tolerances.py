"""The forward-scan tolerance rule of round 3, shared by the parity tests.

North star: the MFMA scan matches the reference's scan within 1e-3 (relative L2) of ARITHMETIC error -- on top of the one
rounding of the bf16 output itself (Q_BF16: no bf16-output kernel can go below it), hence every bound has the form
sqrt(arith^2 + q^2) with q measured on the oracle's own output.

The reference's scan (mamba_ssm 2.2.2, absent from /root/reference) is itself not exact: with 16-bit activations it rounds
three tl.dot operands to the activation dtype (oracle.ssd_ref_chunked(..., emulate_upstream_rounding=True)).
Its own arithmetic error against the fp64 recurrence is 0.4e-3 .. 1.7e-3 depending on the head's decay (heads whose output
is mostly intra-chunk suffer the bf16 C.B^T operand, slow-decay heads the bf16 chunk states), so "within 1e-3 of the
reference" is not a fixed distance from the exact answer.  The rule the tests apply per (batch, head) slice:

    arith(ours vs fp64)  <=  max(ARITH_BUDGET, 1.05 * arith(upstream-rounding oracle at the reference's chunk size 256 vs fp64))

i.e. inside the north-star budget wherever the reference pipeline itself is, and never further from the exact result than
the reference pipeline where its own rounding points exceed the budget.

Two more rules live here and nowhere else (round 6):

  * PRECISE (OmkSsdFwd.flags & OMK_SSD_PRECISE, ssd_combined.scan_options(precise=True)): the BARE 1e-3 on every head -- no
    reference-relative term.  tests/test_configs_gpu.py asserts it on all ten slices of the production shape, slow heads included.
  * the TRAINING instantiation of the default kernel (no final state kept, window-state images saved: what Stage2Step and
    bench.py launch) enters the scaled operand of the state update as ONE bf16 value, like the reference pipeline's chunk-state
    kernel, where the keep-final instantiation uses hi + lo.  It therefore rounds at exactly the reference's two points (that
    operand, and the state that meets C) and ties with it on state-dominated heads: measured 0.94 .. 1.13 x the upstream-rounding
    oracle's own error on heads with A dt0 < 0.01, 0.2 .. 0.6 x elsewhere.  A tie cannot be asserted at 1.05 x of a quantity that
    is itself one draw of a rounding error, so its bound is TRAIN_FACTOR = 1.15 x; callers who need more take KHILO (+ 15 %) or
    PRECISE (+ 37 %) per call (profiles/r06_precise.txt)."""
import math

import torch

import oracle as O

Q_BF16 = 1.65e-3
ARITH_BUDGET = 1e-3
TRAIN_FACTOR = 1.15


def training_budget(upstream_arith):
    """y arithmetic bound of the training instantiation (module docstring)."""
    return max(ARITH_BUDGET, TRAIN_FACTOR * upstream_arith)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def arith_part(e, q):
    return math.sqrt(max(e * e - q * q, 0.0))


def forward_budget(x, dt, A, Bm, Cm, return_upstream=False, **kw):
    """Returns (y_exact fp64 unrounded, final_exact, budget_y, budget_final, (upstream arithmetic error of y, of the final state))
    for one slice of inputs; kw: D, z, dt_bias, initial_states, dt_softplus, dt_limit.  return_upstream: a sixth element, the
    (unrounded) output and final state of the upstream-rounding oracle, for the DIRECT distance (direct_bound below)."""
    kw = dict(kw, return_final_states=True, round_output=False)
    y64, f64 = O.ssd_ref_chunked(x, dt, A, Bm, Cm, 64, compute_dtype=torch.float64, **kw)
    yu, fu = O.ssd_ref_chunked(x, dt, A, Bm, Cm, 256, emulate_upstream_rounding=True, **kw)
    eu, ef = rel(yu, y64), rel(fu, f64)
    r = (y64, f64, max(ARITH_BUDGET, 1.05 * eu), max(ARITH_BUDGET, 1.05 * ef), (eu, ef))
    return r + ((yu, fu),) if return_upstream else r


def direct_bound(own_arith_budget, upstream_arith, q=0.0):
    """Round 4: the distance of this kernel's output to the upstream-rounding oracle ITSELF (not each one's
    distance to the exact result).  Two independent arithmetic errors a (ours, <= own_arith_budget) and u (the reference
    pipeline's own, measured) are at most a + u apart and sqrt(a^2 + u^2) apart when uncorrelated; the tests assert the
    uncorrelated form with 10 % slack, on top of the quantisation q of a bf16 output compared with an unrounded one.  Where the
    reference pipeline's own error is small (u << 1e-3) this IS the north-star 1e-3; where u ~ 1.7e-3 no kernel that does not
    reproduce upstream's rounding errors bit for bit can be closer to it than u."""
    return math.sqrt(q * q + 1.21 * (own_arith_budget ** 2 + upstream_arith ** 2))
