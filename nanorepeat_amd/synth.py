"""Seeded synthetic workloads for the repeat-size scoring path.

All generators are deterministic in `seed`.  Reads are the already-trimmed *core*
sequences the 1D path receives (last `flank` bp of the left flank + unit * k_true +
first `flank` bp of the right flank, through a per-base error channel), or full amplicon
reads for the joint path.
"""
import numpy as np

SEED = 20260116
_BASES = np.frombuffer(b"ACGT", dtype=np.uint8)

# substitution / insertion / deletion rates (cf. nanoRepeat_bam.py:694-701 totals)
ERROR_MODELS = {
    "ont": (0.025, 0.015, 0.03),
    "ont_q20": (0.010, 0.007, 0.013),
    "hifi": (0.003, 0.007, 0.010),
    "none": (0.0, 0.0, 0.0),
}


def rand_seq(rng, n):
    return _BASES[rng.integers(0, 4, size=n)].tobytes().decode()


def rand_unit(rng, m):
    """Random motif of length m that is not a homopolymer."""
    while True:
        u = rand_seq(rng, m)
        if len(set(u)) > 1 or m == 1:
            return u


def apply_errors(rng, seq, model):
    """Per-base channel: delete, else maybe substitute; then maybe insert a uniform base."""
    sub, ins, dele = ERROR_MODELS[model] if isinstance(model, str) else model
    a = np.frombuffer(seq.encode(), dtype=np.uint8)
    n = len(a)
    if n == 0 or (sub == 0 and ins == 0 and dele == 0):
        return seq
    u = rng.random(n)
    keep = u >= dele
    do_sub = keep & (u < dele + sub)
    b = a.copy()
    if do_sub.any():
        # substitute with one of the three other bases
        idx = np.searchsorted(_BASES, b[do_sub])
        b[do_sub] = _BASES[(idx + rng.integers(1, 4, size=int(do_sub.sum()))) % 4]
    do_ins = rng.random(n) < ins
    out = np.empty(2 * n, dtype=np.uint8)
    mask = np.zeros(2 * n, dtype=bool)
    out[0::2] = b
    mask[0::2] = keep
    out[1::2] = _BASES[rng.integers(0, 4, size=n)]
    mask[1::2] = do_ins
    return out[mask].tobytes().decode()


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTNacgtn", "TGCANtgcan"))


def reference_window(r2, fast_mode=False):
    """k window of round3_align (nanoRepeat_bam.py:463-472), pinned by tests/golden."""
    buffer = max(15, int(r2 * 0.05))
    if buffer > 150:
        buffer = 150
    if fast_mode:
        buffer = 15
    kmax = int(r2 + buffer)
    kmin = int(r2 - buffer)
    if kmin < 0:
        kmin = 0
    return kmin, kmax


def make_1d(n_reads, unit, alleles, model="ont", kwin=None, flank=100, anchor=1000, seed=SEED,
            fast_mode=False, rng=None):
    """One region, diploid (or any list of) alleles.  kwin=(kmin,kmax) fixes every read's
    window; otherwise the reference window rule is applied to r2 = k_true + N(0,1)."""
    rng = rng or np.random.default_rng(seed)
    left, right = rand_seq(rng, anchor), rand_seq(rng, anchor)
    reads, kt = [], []
    kmin = np.zeros(n_reads, np.int32)
    kmax = np.zeros(n_reads, np.int32)
    for i in range(n_reads):
        k = int(alleles[int(rng.integers(0, len(alleles)))])
        core = left[len(left) - flank:] + unit * k + right[:flank]
        reads.append(apply_errors(rng, core, model))
        kt.append(k)
        if kwin is not None:
            kmin[i], kmax[i] = kwin
        else:
            r2 = max(0.0, k + float(rng.normal(0.0, 1.0)))
            kmin[i], kmax[i] = reference_window(r2, fast_mode)
    return dict(regions=[(left, unit, right)], reads=reads, kmin=kmin, kmax=kmax,
                read_region=None, k_true=np.array(kt, np.int32))


def apply_errors_batch(rng, seqs, model):
    """apply_errors over many sequences in one vectorised pass (same channel, its own draw order)."""
    sub, ins, dele = ERROR_MODELS[model] if isinstance(model, str) else model
    lens = np.fromiter((len(x) for x in seqs), np.int64, len(seqs))
    a = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    n = len(a)
    owner = np.repeat(np.arange(len(seqs)), lens)
    u = rng.random(n)
    keep = u >= dele
    do_sub = keep & (u < dele + sub)
    b = a.copy()
    if do_sub.any():
        idx = np.searchsorted(_BASES, b[do_sub])
        b[do_sub] = _BASES[(idx + rng.integers(1, 4, size=int(do_sub.sum()))) % 4]
    do_ins = rng.random(n) < ins
    out = np.empty(2 * n, dtype=np.uint8)
    mask = np.zeros(2 * n, dtype=bool)
    out[0::2] = b
    mask[0::2] = keep
    out[1::2] = _BASES[rng.integers(0, 4, size=n)]
    mask[1::2] = do_ins
    new_len = np.bincount(np.repeat(owner, 2)[mask], minlength=len(seqs))
    text = out[mask].tobytes().decode()
    off = np.r_[0, np.cumsum(new_len)]
    return [text[off[i]:off[i + 1]] for i in range(len(seqs))]


def make_1d_batch(n_reads, unit, alleles, model, rng, flank=100, anchor=1000, fast_mode=False):
    """make_1d with the reference window rule, generated in one vectorised pass (config 4 needs a
    million reads)."""
    left, right = rand_seq(rng, anchor), rand_seq(rng, anchor)
    kt = np.asarray(alleles, np.int32)[rng.integers(0, len(alleles), size=n_reads)]
    cores = {int(k): left[len(left) - flank:] + unit * int(k) + right[:flank] for k in set(kt.tolist())}
    reads = apply_errors_batch(rng, [cores[int(k)] for k in kt], model)
    r2 = np.maximum(0.0, kt + rng.normal(0.0, 1.0, size=n_reads))
    buf = np.full(n_reads, 15.0) if fast_mode else np.clip(np.floor(r2 * 0.05), 15, 150)    # reference_window, vectorised
    kmin = np.maximum(np.trunc(r2 - buf), 0).astype(np.int32)
    kmax = np.trunc(r2 + buf).astype(np.int32)
    return dict(regions=[(left, unit, right)], reads=reads, kmin=kmin, kmax=kmax, read_region=None, k_true=kt)


def config2(n_reads=10000, seed=SEED):
    """BASELINE config 2: 10k ONT-error reads, motif TATTG, k in [5,200], alleles 40/150."""
    return make_1d(n_reads, "TATTG", (40, 150), "ont", kwin=(5, 200), seed=seed)


def config5(n_reads=1000, seed=SEED):
    """BASELINE config 5: HiFi model, 5 bp motif, k in [5,500] wide sweep, alleles 60/420."""
    return make_1d(n_reads, "TATTG", (60, 420), "hifi", kwin=(5, 500), seed=seed)


def config4_region(g, seed=SEED):
    """Descriptor of region g of config 4 (no reads): motif, the two alleles and the generator that
    continues into the flanks and reads, so that any rank can materialise any subset of regions."""
    rng = np.random.default_rng([seed, g])
    m = int(rng.integers(3, 7))
    unit = rand_unit(rng, m)
    alleles = (int(rng.integers(10, 121)), int(rng.integers(10, 121)))
    return dict(m=m, unit=unit, alleles=alleles, rng=rng)


def config4_region_cost(desc, reads_per_region, flank=100, anchor=1000):
    """Expected executed cells of a region (what a shard costs; see dist.executed_cells), from its
    descriptor alone: error-free core length and the reference window around each allele."""
    from .dist import padded_rows
    cost = 0
    for a in desc["alleles"]:
        q = 2 * flank + desc["m"] * a
        cost += int(padded_rows(q, desc["m"])) * (2 * anchor + desc["m"] * reference_window(float(a))[1] + 31 * (desc["m"] + 1))
    return cost * reads_per_region // len(desc["alleles"])


def config4(n_regions=1000, reads_per_region=1000, seed=SEED, only=None):
    """BASELINE config 4: many regions, mixed 3-6 bp motifs, reference window rule.  `only`: the
    region numbers to materialise (a rank's shard); read_region then indexes the returned list and
    `region_id` / `read_id` give the global numbering (read_id = region * reads_per_region + i)."""
    ids = range(n_regions) if only is None else sorted(int(g) for g in only)
    regions, reads, rr, kt, gid, rid = [], [], [], [], [], []
    kmins, kmaxs = [], []
    for j, g in enumerate(ids):
        desc = config4_region(g, seed)
        d = make_1d_batch(reads_per_region, desc["unit"], desc["alleles"], "ont_q20", desc["rng"])
        regions.append(d["regions"][0])
        reads += d["reads"]
        rr.append(np.full(reads_per_region, j, np.int32))
        gid.append(g)
        rid.append(g * reads_per_region + np.arange(reads_per_region, dtype=np.int64))
        kmins.append(d["kmin"]); kmaxs.append(d["kmax"]); kt.append(d["k_true"])
    cat = lambda v, dt: np.concatenate(v) if v else np.zeros(0, dt)
    return dict(regions=regions, reads=reads, kmin=cat(kmins, np.int32), kmax=cat(kmaxs, np.int32),
                read_region=cat(rr, np.int32), k_true=cat(kt, np.int32),
                region_id=np.array(gid, np.int64), read_id=cat(rid, np.int64))


def make_joint(n_reads, unit1="CAG", unit2="CCG", mid="CAACAGCCGCCAC",
               alleles=((17, 10), (55, 7)), weights=(0.46, 0.54), model="ont", read_len=1200,
               read_sd=100, anchor=1000, seed=SEED, minus_frac=0.5):
    """BASELINE config 3 shape: HTT-like joint region, full amplicon reads of either strand,
    per-read round-1 ranges [max(0,k-20), k+5) on both axes (nanoRepeat_joint.py:620-637)."""
    rng = np.random.default_rng(seed)
    left, right = rand_seq(rng, anchor), rand_seq(rng, anchor)
    reads, strands, truth, r1, r2 = [], [], [], [], []
    w = np.asarray(weights, float) / np.sum(weights)
    for _ in range(n_reads):
        a = int(rng.choice(len(alleles), p=w))
        k1, k2 = alleles[a]
        core_len = len(unit1) * k1 + len(mid) + len(unit2) * k2
        total = max(core_len + 200, int(rng.normal(read_len, read_sd)))
        fl = (total - core_len) // 2
        fl = min(fl, anchor)
        s = left[len(left) - fl:] + unit1 * k1 + mid + unit2 * k2 + right[:fl]
        s = apply_errors(rng, s, model)
        st = 1
        if rng.random() < minus_frac:
            s = revcomp(s); st = -1
        reads.append(s); strands.append(st); truth.append((k1, k2))
        r1.append((max(0, k1 - 20), k1 + 5))
        r2.append((max(0, k2 - 20), k2 + 5))
    return dict(region=(left, unit1, mid, unit2, right), reads=reads,
                strand=np.array(strands, np.int8), truth=np.array(truth, np.int32),
                range1=np.array(r1, np.int32), range2=np.array(r2, np.int32))


def config3(n_reads=5000, seed=SEED):
    """BASELINE config 3: HTT amplicon joint CAG+CCG quantification, 5k reads."""
    return make_joint(n_reads, seed=seed)
