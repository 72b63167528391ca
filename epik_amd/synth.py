"""Seeded synthetic trees, phylo-k-mer databases and reads.

No real database is available offline (D652 / 16S are downloaded by the
reference's README from another repository, README.md:66-67), so every test and
the benchmark run on the synthetic model fixed in SURVEY.md 8(d) / BASELINE.md 3:

* tree: random rooted binary tree, `n_leaves` leaves => N = 2*n_leaves - 1 nodes,
  branch lengths Exp(mean 0.05), post-order ids 0..N-1 (root = N-1);
* DB: each of the sigma^k keys present with probability `p_present`; list length
  1 + min(N-1, floor(LogNormal(3.0, 1.5))); branches = a contiguous post-order
  run starting at a uniform node (clade locality), distinct, ascending; scores
  log10(U(threshold, 1)) as float32;
* reads: uniform random over the alphabet, fixed length.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import alphabet

#: numpy mirror of `i2l::pkdb_value {branch, score}` (8 bytes, main.cpp:257)
PKDB_VALUE = np.dtype([("branch", np.uint32), ("score", np.float32)])


@dataclass
class SynthTree:
    """A rooted binary tree in post-order numbering.

    `subtree_num_nodes` / `subtree_total_length` mirror `db.tree_index()[i]`
    (place.cpp:113-114): nodes in the subtree rooted at i (i included) and the
    summed branch lengths strictly below i."""

    parent: np.ndarray            # int64[N], -1 for the root
    children: np.ndarray          # int64[N,2], -1 for leaves
    branch_length: np.ndarray     # float64[N]
    labels: list
    subtree_num_nodes: np.ndarray = field(default=None)
    subtree_total_length: np.ndarray = field(default=None)

    @property
    def num_nodes(self) -> int:
        return int(self.parent.shape[0])

    def newick(self, jplace: bool = False) -> str:
        """Newick string; with jplace=True edges carry `{postorder_id}` as
        `i2l::io::to_newick(tree, true)` does (main.cpp:296-297; syntax assumed)."""
        out = []

        def fmt(i: int) -> str:
            s = f"{self.labels[i]}:{self.branch_length[i]:.17g}"
            if jplace:
                s += "{%d}" % i
            return s

        # iterative post-order emit
        root = self.num_nodes - 1
        stack = [(root, 0)]
        while stack:
            node, state = stack.pop()
            l, r = self.children[node]
            if l < 0:
                out.append(fmt(node))
                continue
            if state == 0:
                out.append("(")
                stack.append((node, 1))
                stack.append((int(l), 0))
            elif state == 1:
                out.append(",")
                stack.append((node, 2))
                stack.append((int(r), 0))
            else:
                out.append(")")
                out.append(fmt(node))
        return "".join(out) + ";"


def make_tree(n_leaves: int, seed: int = 42, mean_branch_length: float = 0.05) -> SynthTree:
    """Random rooted binary tree by random joins, renumbered in post-order."""
    if n_leaves < 1:
        raise ValueError("n_leaves must be >= 1")
    rng = np.random.default_rng(seed)
    n = 2 * n_leaves - 1
    # build with temporary ids: leaves 0..n_leaves-1, internal afterwards
    tmp_children = -np.ones((n, 2), dtype=np.int64)
    active = list(range(n_leaves))
    nxt = n_leaves
    while len(active) > 1:
        i, j = rng.choice(len(active), size=2, replace=False)
        a, b = active[i], active[j]
        tmp_children[nxt] = (a, b)
        for idx in sorted((i, j), reverse=True):
            active.pop(idx)
        active.append(nxt)
        nxt += 1
    root = active[0]
    # post-order renumbering
    order = []
    stack = [(root, False)]
    while stack:
        node, done = stack.pop()
        if done or tmp_children[node, 0] < 0:
            order.append(node)
            continue
        stack.append((node, True))
        stack.append((int(tmp_children[node, 1]), False))
        stack.append((int(tmp_children[node, 0]), False))
    new_id = np.empty(n, dtype=np.int64)
    new_id[np.array(order)] = np.arange(n)
    children = -np.ones((n, 2), dtype=np.int64)
    parent = -np.ones(n, dtype=np.int64)
    labels = [""] * n
    for old in range(n):
        ni = new_id[old]
        if tmp_children[old, 0] >= 0:
            l, r = new_id[tmp_children[old, 0]], new_id[tmp_children[old, 1]]
            children[ni] = (l, r)
            parent[l] = ni
            parent[r] = ni
        else:
            labels[ni] = f"t{old}"
    branch_length = rng.exponential(mean_branch_length, size=n)
    branch_length[n - 1] = 0.0  # root
    tree = SynthTree(parent=parent, children=children, branch_length=branch_length, labels=labels)
    num = np.ones(n, dtype=np.int64)
    tot = np.zeros(n, dtype=np.float64)
    for i in range(n):  # post-order: children precede parents
        l, r = children[i]
        if l >= 0:
            num[i] += num[l] + num[r]
            tot[i] += tot[l] + tot[r] + branch_length[l] + branch_length[r]
    tree.subtree_num_nodes = num
    tree.subtree_total_length = tot
    return tree


@dataclass
class SynthDB:
    """A phylo-k-mer database in CSR form: key -> values[offsets[key]:offsets[key+1]]."""

    states: str
    kmer_size: int
    omega: float
    num_branches: int
    offsets: np.ndarray   # uint64[sigma^k + 1]
    values: np.ndarray    # PKDB_VALUE[num_entries]
    threshold: np.float32 = None
    log_threshold: np.float32 = None
    total_entries: int = None     # postings of the whole database when this object holds one shard of it
    keys: np.ndarray = None       # the sparse form: uint32 ascending codes that have a list, offsets[len(keys) + 1]
    shard: tuple = None           # (g, G): this object holds the lists of the codes with code % G == g only

    def __post_init__(self):
        sigma = alphabet.alphabet_size(self.states)
        if self.threshold is None:
            self.threshold = alphabet.score_threshold(self.omega, self.kmer_size, sigma)
        if self.log_threshold is None:
            self.log_threshold = alphabet.log_threshold(self.threshold)

    @property
    def alphabet_size(self) -> int:
        return alphabet.alphabet_size(self.states)

    @property
    def num_keys(self) -> int:
        return self.alphabet_size ** self.kmer_size if self.keys is not None else int(self.offsets.shape[0] - 1)

    def densified(self) -> "SynthDB":
        """The same database with an offset per possible code (8 bytes each: what the CPU oracle indexes by)."""
        if self.keys is None:
            return self
        lens = np.zeros(self.num_keys + 1, dtype=np.uint64)
        lens[self.keys.astype(np.int64) + 1] = np.diff(self.offsets.astype(np.int64)).astype(np.uint64)
        np.cumsum(lens, out=lens)
        return SynthDB(states=self.states, kmer_size=self.kmer_size, omega=self.omega, num_branches=self.num_branches,
                       offsets=lens, values=self.values, threshold=self.threshold, log_threshold=self.log_threshold,
                       total_entries=self.total_entries)

    @property
    def num_entries(self) -> int:
        return int(self.values.shape[0])


def make_db(num_branches: int, states: str = "nucl", kmer_size: int = 10, omega: float = 1.5,
            p_present: float = 0.6, seed: int = 43, lognormal=(3.0, 1.5),
            scattered: bool = False, shard=None) -> SynthDB:
    """Synthetic DB of SURVEY.md 8(d).  `scattered=True` replaces the contiguous
    branch run by a sorted random subset (stress for the LDS scatter-add).
    `shard=(g, G)`: only the lists of the codes with code % G == g are generated (every other code an
    empty list) -- a rank of a k-mer-space-sharded job never holds the rest; which codes exist and how
    long their lists are is the same draw as without `shard`, the postings are the shard's own."""
    sigma = alphabet.alphabet_size(states)
    num_keys = sigma ** kmer_size
    rng = np.random.default_rng(seed)
    threshold = alphabet.score_threshold(omega, kmer_size, sigma)
    present = rng.random(num_keys) < p_present
    n_present = int(present.sum())
    raw = np.floor(rng.lognormal(lognormal[0], lognormal[1], size=n_present))
    lengths = (1 + np.minimum(num_branches - 1, raw)).astype(np.int64)
    total_all = int(lengths.sum())
    if shard is not None:
        g, count = shard
        keys = np.nonzero(present)[0]
        mine = keys % count == g
        present = np.zeros(num_keys, dtype=bool)
        present[keys[mine]] = True
        lengths = lengths[mine]
        n_present = int(mine.sum())
        rng = np.random.default_rng([seed, 1000 + g])
    lens_all = np.zeros(num_keys, dtype=np.int64)
    lens_all[present] = lengths
    offsets = np.zeros(num_keys + 1, dtype=np.uint64)
    np.cumsum(lens_all, out=offsets[1:].view(np.int64))
    total = int(offsets[-1])
    values = np.empty(total, dtype=PKDB_VALUE)
    list_id = np.repeat(np.arange(n_present, dtype=np.int64), lengths)
    list_start = np.cumsum(lengths) - lengths
    within = np.arange(total, dtype=np.int64) - list_start[list_id]
    if scattered:
        # `len` distinct branches per list: (start + j * step) mod N with step coprime
        # to N, then sorted ascending within each list
        start = rng.integers(0, num_branches, size=n_present)
        step = rng.integers(1, max(2, num_branches), size=n_present)
        g = np.gcd(step, num_branches)
        step = np.where(g == 1, step, 1)
        br = (start[list_id] + within * step[list_id]) % num_branches
        order = np.lexsort((br, list_id))
        br = br[order]
    else:
        start = (rng.random(n_present) * (num_branches - lengths + 1)).astype(np.int64)
        br = start[list_id] + within
    values["branch"] = br.astype(np.uint32)
    u = rng.random(total)
    prob = float(threshold) + u * (1.0 - float(threshold))
    values["score"] = np.log10(prob).astype(np.float32)
    db = SynthDB(states=states, kmer_size=kmer_size, omega=omega, num_branches=num_branches,
                 offsets=offsets, values=values, threshold=threshold)
    db.total_entries = total_all  # of the whole database, whatever the shard
    if shard is not None and shard[1] > 1:
        db.shard = (int(shard[0]), int(shard[1]))
    return db


def make_reads(n_reads: int, length: int, states: str = "nucl", seed: int = 44):
    """Uniform random reads of fixed length.  Returns (bytes uint8[n*length], offsets uint64[n+1])."""
    rng = np.random.default_rng(seed)
    chars = np.frombuffer(alphabet.state_chars(states).encode(), dtype=np.uint8)
    seqs = chars[rng.integers(0, len(chars), size=n_reads * length, dtype=np.uint8)]
    offsets = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(length)
    return seqs, offsets


def pack_reads(reads) -> tuple:
    """Concatenates an iterable of str/bytes reads -> (uint8 buffer, uint64 offsets[n+1])."""
    bufs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    offsets = np.zeros(len(bufs) + 1, dtype=np.uint64)
    if bufs:
        offsets[1:] = np.cumsum([len(b) for b in bufs], dtype=np.uint64)
    data = np.frombuffer(b"".join(bufs), dtype=np.uint8).copy() if bufs else np.zeros(0, np.uint8)
    return data, offsets


def make_sparse_db(num_branches: int, states: str = "amino", kmer_size: int = 7, omega: float = 1.5,
                   p_present: float = 0.0026, seed: int = 43, lognormal=(3.0, 1.5), dense: bool = True) -> SynthDB:
    """The database model of `make_db` for a key space too large to draw one uniform number per code
    (amino k = 7: 1.28 G codes): the present codes are drawn directly (Binomial count, distinct uniform
    codes), everything else is the same.  Peak host memory = the uint64 offsets array (8 B per code)."""
    sigma = alphabet.alphabet_size(states)
    num_keys = sigma ** kmer_size
    rng = np.random.default_rng(seed)
    threshold = alphabet.score_threshold(omega, kmer_size, sigma)
    n_present = int(rng.binomial(num_keys, p_present))
    keys = np.unique(rng.integers(0, num_keys, size=int(n_present * 1.02) + 16, dtype=np.int64))
    keys = np.sort(rng.permutation(keys)[:n_present])
    n_present = int(keys.shape[0])
    raw = np.floor(rng.lognormal(lognormal[0], lognormal[1], size=n_present))
    lengths = (1 + np.minimum(num_branches - 1, raw)).astype(np.int64)
    if dense:
        offsets = np.zeros(num_keys + 1, dtype=np.uint64)
        offsets[keys + 1] = lengths.astype(np.uint64)
        np.cumsum(offsets, out=offsets)
    else:
        offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint64)
    total = int(offsets[-1])
    values = np.empty(total, dtype=PKDB_VALUE)
    list_id = np.repeat(np.arange(n_present, dtype=np.int64), lengths)
    list_start = np.cumsum(lengths) - lengths
    within = np.arange(total, dtype=np.int64) - list_start[list_id]
    start = (rng.random(n_present) * (num_branches - lengths + 1)).astype(np.int64)
    values["branch"] = (start[list_id] + within).astype(np.uint32)
    prob = float(threshold) + rng.random(total) * (1.0 - float(threshold))
    values["score"] = np.log10(prob).astype(np.float32)
    return SynthDB(states=states, kmer_size=kmer_size, omega=omega, num_branches=num_branches,
                   offsets=offsets, values=values, threshold=threshold, keys=None if dense else keys.astype(np.uint32))


def reads_hitting(db: SynthDB, n_reads: int, length: int, hit_rate: float, seed: int = 45, dirty: str = ""):
    """Reads for a sparse database: uniform random reads would find next to nothing in it, so every
    read is a chain of k-mers that ARE in the database (drawn among the present codes) joined by random
    letters, about `hit_rate` of its positions starting a planted k-mer; `dirty` letters (ambiguous /
    invalid characters) replace a few positions of every fourth read."""
    rng = np.random.default_rng(seed)
    sigma, k = db.alphabet_size, db.kmer_size
    chars = np.frombuffer(alphabet.state_chars(db.states).encode(), dtype=np.uint8)
    lens = np.diff(db.offsets.view(np.int64)) if db.keys is None and db.num_keys <= (1 << 26) else None
    if db.keys is not None:  # the sparse form: the present codes are listed
        present = db.keys.astype(np.int64)
    elif lens is not None:
        present = np.nonzero(lens)[0]
    else:  # large key space: sample codes until enough present ones are found
        present = np.zeros(0, dtype=np.int64)
        while present.shape[0] < 200_000:
            cand = rng.integers(0, db.num_keys, size=4_000_000, dtype=np.int64)
            present = np.concatenate([present, cand[db.offsets[cand + 1] > db.offsets[cand]]])
    seqs = chars[rng.integers(0, sigma, size=(n_reads, length))]
    n_plant = max(1, int(length * hit_rate / k))
    pos = rng.integers(0, length - k + 1, size=(n_reads, n_plant))
    code = present[rng.integers(0, present.shape[0], size=(n_reads, n_plant))]
    for j in range(k):  # letter j of the planted k-mer, first character most significant
        digit = (code // (sigma ** (k - 1 - j))) % sigma
        np.put_along_axis(seqs, pos + j, chars[digit], axis=1)
    if dirty:
        d = np.frombuffer(dirty.encode(), dtype=np.uint8)
        rows = np.arange(0, n_reads, 4)
        for _ in range(3):
            seqs[rows, rng.integers(0, length, size=rows.shape[0])] = d[rng.integers(0, d.shape[0], size=rows.shape[0])]
    offsets = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(length)
    return seqs.reshape(-1).copy(), offsets


def make_clade_db(num_branches: int, kmer_size: int = 10, omega: float = 1.5, n_refs: int = 500, ref_length: int = 1500,
                  width=(3.4, 0.6), seed: int = 47):
    """A database shaped like one built from reference sequences (what a real phylo-k-mer database is): `n_refs`
    random nucleotide references, each at home in a clade of the tree -- a run of branches around a centre --, and
    every k-mer of a reference carries a list over (most of) that clade: the lists of CONSECUTIVE k-mers of a
    reference cover nearly the same branches, their scores highest at the clade's centre.  make_db() draws every
    list's position independently, so the k-mers of a read land all over the tree; here a read cut from a reference
    (make_clade_reads) adds k-mer after k-mer into the same few dozen rows -- consecutive chunks of the stream
    update the SAME LDS rows, and a row's count climbs to the number of k-mers of the read.
    Returns (SynthDB, refs uint8[n_refs, ref_length] of states 0..3, centres)."""
    assert kmer_size <= 15
    rng = np.random.default_rng(seed)
    sigma = 4
    num_keys = sigma ** kmer_size
    threshold = alphabet.score_threshold(omega, kmer_size, sigma)
    refs = rng.integers(0, sigma, size=(n_refs, ref_length), dtype=np.uint8)
    centres = rng.integers(0, num_branches, size=n_refs)
    n_pos = ref_length - kmer_size + 1
    codes = np.zeros((n_refs, n_pos), dtype=np.int64)
    for j in range(kmer_size):
        codes = codes * sigma + refs[:, j:j + n_pos]
    # a code met in several places keeps the first one (reference order)
    flat = codes.reshape(-1)
    uniq, first = np.unique(flat, return_index=True)
    ref_of = first // n_pos
    # the list of a k-mer: centre +- a few branches of jitter, a width of its own around the clade's
    clade_w = np.maximum(2, rng.lognormal(width[0], width[1], size=n_refs)).astype(np.int64)
    w = np.maximum(1, (clade_w[ref_of] * rng.uniform(0.6, 1.2, size=uniq.shape[0])).astype(np.int64))
    w = np.minimum(w, num_branches)
    mid = centres[ref_of] + rng.integers(-3, 4, size=uniq.shape[0])
    start = np.clip(mid - w // 2, 0, num_branches - w)
    offsets = np.zeros(num_keys + 1, dtype=np.uint64)
    lens = np.zeros(num_keys, dtype=np.int64)
    lens[uniq] = w
    np.cumsum(lens, out=offsets[1:].view(np.int64))
    total = int(offsets[-1])
    values = np.empty(total, dtype=PKDB_VALUE)
    list_id = np.repeat(np.arange(uniq.shape[0], dtype=np.int64), w)   # uniq is ascending: CSR order
    list_start = np.cumsum(w) - w
    within = np.arange(total, dtype=np.int64) - list_start[list_id]
    br = start[list_id] + within
    values["branch"] = br.astype(np.uint32)
    # probability: near 1 at the clade's centre, falling towards the threshold at its edges (+ noise)
    dist = np.abs(br - centres[ref_of][list_id]).astype(np.float64) / np.maximum(1, clade_w[ref_of][list_id])
    closeness = np.clip(1.0 - dist + rng.normal(0.0, 0.15, size=total), 0.02, 1.0)
    prob = float(threshold) + closeness * (0.9 - float(threshold))
    values["score"] = np.log10(prob).astype(np.float32)
    db = SynthDB(states="nucl", kmer_size=kmer_size, omega=omega, num_branches=num_branches, offsets=offsets,
                 values=values, threshold=threshold)
    db.total_entries = total
    return db, refs, centres


def make_clade_reads(refs: np.ndarray, n_reads: int, length: int, substitutions: float = 0.01, seed: int = 48):
    """Reads cut from the references of make_clade_db at random places, `substitutions` of their letters replaced."""
    rng = np.random.default_rng(seed)
    n_refs, ref_length = refs.shape
    chars = np.frombuffer(alphabet.state_chars("nucl").encode(), dtype=np.uint8)
    which = rng.integers(0, n_refs, size=n_reads)
    at = rng.integers(0, ref_length - length + 1, size=n_reads)
    idx = at[:, None] + np.arange(length)[None, :]
    states = refs[which[:, None], idx]
    flip = rng.random(size=states.shape) < substitutions
    states = np.where(flip, rng.integers(0, 4, size=states.shape, dtype=np.uint8), states)
    seqs = chars[states]
    offsets = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(length)
    return np.ascontiguousarray(seqs.reshape(-1)), offsets
