"""the on-device sampler (omk_sample, csrc/sample.hip) against the semantics of the reference's
`sample` (/root/reference/models/stage2/generation.py:87-121 with the top-p filter of :64-76):
  * top_k == 1: the ids of torch.argmax, bit-exact;
  * 1 < top_k <= 64 with temperature and top-p: the candidate set is exactly the reference's (torch.topk + the ascending
    cumulative-probability cut), and the empirical distribution of 4096 draws (different Philox streams) matches the reference's
    probabilities by chi-square; a fixed (seed, counter) reproduces its ids;
  * ties at the k-th logit (16-bit logits tie often) resolve towards the lowest indices, deterministically."""
import math

import pytest
import torch


def ref_distribution(logits_row, top_k, top_p, temperature):
    """Probabilities over the vocabulary as the reference's sample() draws them (generation.py:94-104)."""
    vals, idx = torch.topk(logits_row.float(), top_k)
    vals = vals / temperature
    if 0.0 < top_p < 1.0:
        sv, si = torch.sort(vals, descending=False)
        remove = sv.softmax(-1).cumsum(-1) <= (1 - top_p)
        vals = vals.masked_fill(remove.scatter(0, si, remove), float("-inf"))
    p = torch.zeros_like(logits_row, dtype=torch.float64)
    p[idx] = torch.softmax(vals.double(), -1)
    return p


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_greedy_equals_argmax(dev, dtype):
    from omnimamba_amd.sampling import sample_device
    g = torch.Generator().manual_seed(0)
    V = 50288 if dev.type == "cuda" else 3000
    logits = torch.randn(5, V, generator=g).to(dtype)
    if dtype == torch.bfloat16:          # make the maximum unique (bf16 rows tie; the kernel takes the lowest index, torch any)
        for b in range(5):
            logits[b, 17 * (b + 1)] = 9.0
    ids = sample_device(logits.to(dev), top_k=1)
    assert torch.equal(ids.cpu(), logits.float().argmax(-1))


@pytest.mark.parametrize("top_k,top_p,temp", [(8, 0.0, 1.0), (20, 0.9, 0.7), (64, 0.6, 1.3), (3, 0.999, 1.0)])
def test_topk_topp_distribution_matches_reference(dev, top_k, top_p, temp):
    from omnimamba_amd.sampling import sample_device
    g = torch.Generator().manual_seed(1)
    V = 16384 if dev.type == "cuda" else 700
    row = torch.randn(V, generator=g) * 2.0
    p_ref = ref_distribution(row, top_k, top_p, temp)
    support = (p_ref > 0).nonzero().squeeze(-1)
    nrow, nlaunch = (256, 16) if dev.type == "cuda" else (48, 14)
    logits = row[None].repeat(nrow, 1).contiguous().to(dev)
    counter = torch.zeros((), dtype=torch.int64, device=dev)
    counts = torch.zeros(V, dtype=torch.float64)
    for i in range(nlaunch):
        counter.fill_(i)
        ids = sample_device(logits, top_k=top_k, top_p=top_p, temperature=temp, seed=1234, step_counter=counter).cpu()
        counts += torch.bincount(ids, minlength=V).double()
    n = nrow * nlaunch
    assert counts.sum() == n
    assert (counts[p_ref == 0] == 0).all(), "a token outside the reference's candidate set was drawn"
    # chi-square over the support (cells with expectation < 5 pooled)
    exp = p_ref[support] * n
    obs = counts[support]
    big = exp >= 5
    chi = (((obs[big] - exp[big]) ** 2) / exp[big]).sum().item()
    dof = int(big.sum().item()) - 1
    if (~big).any():
        e, o = exp[~big].sum().item(), obs[~big].sum().item()
        if e > 0:
            chi += (o - e) ** 2 / e
            dof += 1
    # mean dof, variance 2 dof: five sigma
    assert chi < dof + 5 * math.sqrt(2 * max(dof, 1)) + 5, (chi, dof)
    # reproducible: the same (seed, counter) draws the same ids; another seed does not (64 candidates, many rows)
    counter.fill_(3)
    a = sample_device(logits, top_k=top_k, top_p=top_p, temperature=temp, seed=1234, step_counter=counter)
    b = sample_device(logits, top_k=top_k, top_p=top_p, temperature=temp, seed=1234, step_counter=counter)
    c = sample_device(logits, top_k=top_k, top_p=top_p, temperature=temp, seed=99, step_counter=counter)
    assert torch.equal(a, b)
    if len(support) > 1:
        assert not torch.equal(a, c)


@pytest.mark.parametrize("temp,top_p", [(1.0, 1.0), (0.6, 0.0)])
def test_full_vocabulary_multinomial_matches_reference(dev, temp, top_p):
    """top_k == 0 with top_p outside (0, 1) -- the default arguments of the reference's t2i_generate (omnimamba.py:311) -- is the plain
    multinomial of softmax(logits / T) over the whole vocabulary (generation.py:114-119): chi-square of 8192 / 2048 draws against those
    probabilities, -inf logits never drawn, reproducible per (seed, counter)."""
    from omnimamba_amd.sampling import applies, sample_device
    g = torch.Generator().manual_seed(5)
    V = 16384 if dev.type == "cuda" else 1500
    row = torch.randn(V, generator=g) * 3.0
    row[::7] = float("-inf")
    p_ref = torch.softmax(row.double() / temp, -1)
    nrow, nlaunch = (512, 16) if dev.type == "cuda" else (48, 16)
    logits = row[None].repeat(nrow, 1).contiguous().to(dev)
    assert applies(logits, 0, top_p=top_p) and applies(logits, 0, top_p=0.5) and applies(logits, 0, min_p=0.1, top_p=top_p) and not applies(logits, 0, min_p=1.5)
    counter = torch.zeros((), dtype=torch.int64, device=dev)
    counts = torch.zeros(V, dtype=torch.float64)
    for i in range(nlaunch):
        counter.fill_(i)
        counts += torch.bincount(sample_device(logits, top_k=0, top_p=top_p, temperature=temp, seed=77, step_counter=counter).cpu(), minlength=V).double()
    n = nrow * nlaunch
    assert counts.sum() == n and (counts[p_ref == 0] == 0).all()
    exp = p_ref * n
    big = exp >= 5
    chi = (((counts[big] - exp[big]) ** 2) / exp[big]).sum().item()
    dof = int(big.sum().item()) - 1
    e, o = exp[~big].sum().item(), counts[~big].sum().item()
    if e > 0:
        chi += (o - e) ** 2 / e
        dof += 1
    assert chi < dof + 5 * math.sqrt(2 * max(dof, 1)) + 5, (chi, dof)
    counter.fill_(2)
    a = sample_device(logits, top_k=0, top_p=top_p, temperature=temp, seed=77, step_counter=counter)
    b = sample_device(logits, top_k=0, top_p=top_p, temperature=temp, seed=77, step_counter=counter)
    assert torch.equal(a, b) and not torch.equal(a, sample_device(logits, top_k=0, top_p=top_p, temperature=temp, seed=78, step_counter=counter))


def _ref_full_filtered(row, top_p, min_p, temp):
    """The reference's whole-vocabulary branch (generation.py:107-119) as probabilities over the vocabulary + the mask of removed tokens."""
    row = row.float()
    if min_p > 0.0:
        thr = torch.softmax(row, -1).max() * min_p          # a probability ... compared with the raw logits (:43, :110-111)
        removed = row < thr
        work = row.masked_fill(removed, float("-inf")) / temp
    else:
        work = row / temp
        sv, si = torch.sort(work, descending=False, stable=True)
        rem_sorted = sv.softmax(-1).cumsum(-1) <= (1 - top_p)
        removed = torch.zeros_like(rem_sorted).scatter(0, si, rem_sorted)
        work = work.masked_fill(removed, float("-inf"))
    return torch.softmax(work.double(), -1), removed


@pytest.mark.parametrize("top_p,min_p,temp", [(0.9, 0.0, 1.0), (0.5, 0.0, 0.7), (0.05, 0.0, 1.3), (0.0, 0.02, 1.0), (0.3, 0.2, 0.8)])
def test_full_vocabulary_top_p_and_min_p_filters_match_reference(dev, top_p, min_p, temp):
    """top_k == 0 behind the reference's two filters of that branch (generation.py:107-119), on the device since round 6: the top-p cut over
    ALL tokens (ascending cumulative probability <= 1 - top_p is cut) and min_p (a raw logit under min_p x the largest probability is cut; top_p
    is ignored then).  No draw may land on a token the reference removes (the one token at the boundary of the cumulative sum excepted: the
    reference adds fp32 probabilities in sorted order, the kernel 2^-40 fixed-point masses), and 4096+ draws follow its probabilities."""
    from omnimamba_amd.sampling import applies, sample_device
    g = torch.Generator().manual_seed(11)
    V = 16384 if dev.type == "cuda" else 1200
    row = torch.randn(V, generator=g) * 2.5 + 1.0
    row[::11] = float("-inf")
    p_ref, removed = _ref_full_filtered(row, top_p, min_p, temp)
    nrow, nlaunch = (512, 16) if dev.type == "cuda" else (48, 10)
    logits = row[None].repeat(nrow, 1).contiguous().to(dev)
    assert applies(logits, 0, top_p=top_p, min_p=min_p)
    counter = torch.zeros((), dtype=torch.int64, device=dev)
    counts = torch.zeros(V, dtype=torch.float64)
    for i in range(nlaunch):
        counter.fill_(i)
        counts += torch.bincount(sample_device(logits, top_k=0, top_p=top_p, min_p=min_p, temperature=temp, seed=21, step_counter=counter).cpu(), minlength=V).double()
    n = nrow * nlaunch
    assert counts.sum() == n
    wrong = torch.nonzero((counts > 0) & removed).flatten()
    if min_p > 0.0:
        assert wrong.numel() == 0
    else:   # at most the largest removed token (the boundary of the cumulative sum)
        finite_removed = torch.where(removed & torch.isfinite(row), row, torch.full_like(row, float("-inf")))
        assert wrong.numel() <= 1 and (wrong.numel() == 0 or row[wrong[0]] == finite_removed.max())
    exp = p_ref * n
    big = exp >= 5
    chi = (((counts[big] - exp[big]) ** 2) / exp[big]).sum().item()
    dof = int(big.sum().item()) - 1
    e, o = exp[~big].sum().item(), counts[~big].sum().item()
    if e > 0:
        chi += (o - e) ** 2 / e
        dof += 1
    assert chi < dof + 5 * math.sqrt(2 * max(dof, 1)) + 5, (chi, dof)
    counter.fill_(3)
    a = sample_device(logits, top_k=0, top_p=top_p, min_p=min_p, temperature=temp, seed=21, step_counter=counter)
    assert torch.equal(a, sample_device(logits, top_k=0, top_p=top_p, min_p=min_p, temperature=temp, seed=21, step_counter=counter))


def test_full_vocabulary_top_p_ties_and_empty_min_p(dev):
    """Equal logits at the boundary of the top-p cut are removed in index order (what a stable ascending sort does); a tiny top_p leaves the
    arg max alone; min_p with every logit under the threshold (all-negative logits: the reference's multinomial raises on the NaN row) gives the
    arg max."""
    from omnimamba_amd.sampling import sample_device
    V = 700
    row = torch.full((V,), -3.0)
    row[100:140] = 1.0          # forty equal tokens hold most of the mass: the cut ends inside them
    row[5] = 2.0
    p_ref, removed = _ref_full_filtered(row, 0.6, 0.0, 1.0)
    kept_ties = [i for i in range(100, 140) if not removed[i]]
    assert 0 < len(kept_ties) < 40 and kept_ties == list(range(140 - len(kept_ties), 140))      # the reference cuts the lowest indices
    logits = row[None].repeat(64, 1).contiguous().to(dev)
    counter = torch.zeros((), dtype=torch.int64, device=dev)
    seen = set()
    for i in range(24 if dev.type == "cuda" else 10):
        counter.fill_(i)
        seen |= set(sample_device(logits, top_k=0, top_p=0.6, seed=9, step_counter=counter).cpu().tolist())
    assert seen <= set(kept_ties) | {5} and len(seen & set(kept_ties)) >= len(kept_ties) - 2, (sorted(seen), kept_ties)
    ids = sample_device(logits, top_k=0, top_p=1e-6, seed=9)
    assert (ids.cpu() == 5).all()
    neg = (-torch.rand(3, V) - 0.5).to(dev)
    ids = sample_device(neg, top_k=0, min_p=0.5, seed=1)
    assert torch.equal(ids.cpu(), neg.cpu().argmax(-1))


def test_ties_at_the_threshold_take_the_lowest_indices(dev):
    from omnimamba_amd.sampling import sample_device
    V = 2048
    row = torch.full((V,), -5.0)
    row[100] = 3.0
    tied = [7, 300, 301, 999, 1500, 2000]          # six logits tie for the remaining three slots of top_k = 4
    row[tied] = 1.0
    logits = row[None].repeat(64, 1).bfloat16().to(dev)
    seen = set()
    counter = torch.zeros((), dtype=torch.int64, device=dev)
    for i in range(12):
        counter.fill_(i)
        seen |= set(sample_device(logits, top_k=4, temperature=2.0, seed=5, step_counter=counter).cpu().tolist())
    assert seen <= {100, 7, 300, 301} and {100, 7, 300, 301} <= seen, seen


def test_tiny_top_p_is_greedy_and_argument_checks(dev):
    from omnimamba_amd.sampling import sample_device
    g = torch.Generator().manual_seed(2)
    logits = torch.randn(16, 900, generator=g).to(dev)
    ids = sample_device(logits, top_k=10, top_p=1e-6, seed=3)
    assert torch.equal(ids.cpu(), logits.cpu().argmax(-1))
    with pytest.raises(RuntimeError):
        sample_device(logits, top_k=65)
    with pytest.raises(RuntimeError):
        sample_device(logits, top_k=4, temperature=0.0)


def test_host_sample_calls_advance_the_stream_and_reseeding_reproduces_them(dev):
    """generation.sample() on the device sampler: two calls in a row are independent draws (the reference's torch.multinomial advances
    the global generator), and torch.manual_seed(s) followed by the same calls gives the same ids (advisor finding, round 3: a fixed
    stream position per call / a module counter that re-seeding did not reset)."""
    from omnimamba_amd.generation import sample
    torch.manual_seed(5)
    logits = torch.randn(64, 4096).to(dev)
    torch.manual_seed(77)
    a1 = sample(logits, top_k=40, top_p=0.95, temperature=1.1).cpu()
    a2 = sample(logits, top_k=40, top_p=0.95, temperature=1.1).cpu()
    torch.manual_seed(77)
    b1 = sample(logits, top_k=40, top_p=0.95, temperature=1.1).cpu()
    b2 = sample(logits, top_k=40, top_p=0.95, temperature=1.1).cpu()
    assert torch.equal(a1, b1) and torch.equal(a2, b2)
    assert not torch.equal(a1, a2)
