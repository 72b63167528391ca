"""Second, independently written restatement of EPIK's placement loop (pure
Python + numpy float32 scalars; small cases only).

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle/epik_oracle.h): the
reference has no fixtures and cannot be built here; this file exists so that the
C oracle is checked against a transcription made separately from the same
reference lines, with the reference's own data structures (a key -> list hash map,
dense per-branch arrays, an edge list, std::unordered_set stand-in).

All line numbers are /root/reference/epik/src/epik/place.cpp.
"""
from __future__ import annotations

import math

import numpy as np

f32 = np.float32


class RefShapedPlacer:
    """Mirror of `epik::placer` (place.h:81-140) over a dict database."""

    def __init__(self, db: dict, *, kmer_size: int, alphabet_size: int, num_branches: int,
                 threshold, log_threshold, char_class, keep_at_most: int = 7,
                 keep_factor: float = 0.01):
        self.db = db                      # key -> list[(branch, score float32)]
        self.k = int(kmer_size)
        self.sigma = int(alphabet_size)
        self.n = int(num_branches)
        self.threshold = f32(threshold)   # :87
        self.log_threshold = f32(log_threshold)  # :88
        self.keep_at_most = int(keep_at_most)
        self.keep_factor = float(keep_factor)
        self.char_class = [int(x) for x in char_class]
        # :92-96
        self.scores = [f32(0)] * self.n
        self.scores_amb = [f32(0)] * self.n
        self.counts = [0] * self.n
        self.counts_amb = [0] * self.n
        self.edges: list = []

    # --- i2l::to_kmers<one_ambiguity_policy> (call site :294) -----------------
    def to_kmers(self, seq: bytes):
        """Yields (position, [keys]) for every window with no invalid character
        and at most one ambiguous one; keys in ascending state order."""
        k, sigma = self.k, self.sigma
        for p in range(len(seq) - k + 1):
            digits = []
            amb = []
            ok = True
            for j in range(k):
                cls = self.char_class[seq[p + j]]
                if cls == 0:
                    ok = False
                    break
                states = [s for s in range(sigma) if (cls >> s) & 1]
                if len(states) > 1:
                    amb.append((j, states))
                    digits.append(0)
                else:
                    digits.append(states[0])
            if not ok or len(amb) > 1:
                continue

            def encode(ds):
                key = 0
                for d in ds:
                    key = key * sigma + d
                return key

            if not amb:
                yield p, [encode(digits)]
            else:
                j, states = amb[0]
                keys = []
                for s in states:
                    ds = list(digits)
                    ds[j] = s
                    keys.append(encode(ds))
                yield p, keys

    def to_kmers_positions(self, seq: bytes):
        """`to_kmers` with the resolved state beside every key of an ambiguous window:
        (position, [key]) or (position, [(state, key), ...])."""
        k, sigma = self.k, self.sigma
        for p, keys in self.to_kmers(seq):
            if len(keys) == 1:
                yield p, keys
                continue
            j = next(j for j in range(k) if bin(int(self.char_class[seq[p + j]])).count("1") > 1)
            weight = sigma ** (k - 1 - j)
            yield p, [((key // weight) % sigma, key) for key in keys]

    # --- query_kmers (:278-316) ---------------------------------------------
    def query_kmers(self, seq: bytes):
        exact, ambiguous = [], []
        for _, keys in self.to_kmers(seq):
            if len(keys) == 1:
                hit = self.db.get(keys[0])
                if hit:                      # :301
                    exact.append(hit)
            else:
                for key in keys:             # :308-312, one vector per key
                    ambiguous.append([self.db.get(key)])
        return exact, ambiguous

    # --- place_seq (:320-440) --------------------------------------------------
    def place_seq(self, seq: bytes):
        k = self.k
        num_kmers = len(seq) - k + 1
        for e in self.edges:                 # :335-341
            self.counts[e] = 0
            self.scores[e] = f32(0)
            self.counts_amb[e] = 0
            self.scores_amb[e] = f32(0)
        self.edges = []
        exact, ambiguous = self.query_kmers(seq)
        for hit in exact:                    # :349-371
            for branch, score in hit:
                if self.counts[branch] == 0:
                    self.edges.append(branch)
                self.counts[branch] += 1
                self.scores[branch] = f32(self.scores[branch] + f32(score))
        for amb_result in ambiguous:         # :375-415
            l_amb = []                       # unordered_set; insertion order kept
            for hit in amb_result:
                if hit:
                    for branch, score in hit:
                        if self.counts_amb[branch] == 0:
                            l_amb.append(branch)
                        self.counts_amb[branch] += 1
                        # :391 std::pow(10, float) -> double pow -> float
                        self.scores_amb[branch] = f32(
                            self.scores_amb[branch] + f32(math.pow(10.0, float(f32(score)))))
                    w_size = k               # :395
                    for branch in l_amb:     # :398-411
                        avg = f32(f32(self.scores_amb[branch]
                                      + f32(f32(w_size - self.counts_amb[branch]) * self.threshold))
                                  / f32(w_size))
                        if self.counts[branch] == 0:
                            self.edges.append(branch)
                        self.counts[branch] += 1
                        self.scores[branch] = f32(self.scores[branch] + avg)
        for e in self.edges:                 # :418-422
            self.scores[e] = f32(self.scores[e]
                                 + f32(f32(num_kmers - self.counts[e]) * self.log_threshold))
            self.scores[e] = f32(self.scores[e] / f32(k))
        # :424-438 (distal/pendant are joined by the host, not here)
        return [(e, self.scores[e], self.counts[e]) for e in self.edges]

    # --- sum_scores (:164-184) ---------------------------------------------
    def sum_scores(self, placements, seq_len: int) -> float:
        num_branches = f32(self.n)
        num_placements = f32(len(placements))
        num_kmers = f32(seq_len - self.k + 1)
        kmer_size = f32(self.k)
        expo = f32(f32(num_kmers * self.log_threshold) / kmer_size)
        sum_not_placed = float(f32(num_branches - num_placements)) * math.pow(10.0, float(expo))
        sum_placed = 0.0
        for _, score, _ in placements:
            sum_placed += math.pow(10.0, float(score))
        return sum_not_placed + sum_placed

    # --- select_best_placements (:134-159), ties: (score desc, branch asc) ------
    def select_best(self, placements, num_kmers: int):
        return_size = min(self.keep_at_most, len(placements))
        if return_size == 0:
            return_size = self.keep_at_most
            thr_score = f32(f32(self.log_threshold * f32(num_kmers)) / f32(self.k))
            placements = [(i, thr_score, 0) for i in range(self.keep_at_most)]
        ordered = sorted(placements, key=lambda p: (-float(p[1]), p[0]))
        return ordered[:return_size]

    # --- place(), per-read part (:232-267) -----------------------------------
    def place(self, seq: bytes):
        if len(seq) < self.k:
            return None
        keep_factor = self.keep_factor
        placements = self.place_seq(seq)
        score_sum = self.sum_scores(placements, len(seq))
        num_kmers = len(seq) - self.k + 1
        best = self.select_best(placements, num_kmers)
        rows = []
        for branch, score, count in best:
            if score_sum == 0:
                lwr = 0.0
                keep_factor = 0.0
            else:
                power = math.pow(10.0, float(score))
                lwr = 0.0 if power == 0.0 else power / score_sum
            rows.append([branch, score, lwr, count])
        best_ratio = rows[0][2] if rows else 0.0   # :191
        cut = best_ratio * keep_factor             # :192
        return [r for r in rows if r[2] >= cut]    # :196-197


def dict_db_from_csr(offsets, values) -> dict:
    """key -> [(branch, score)] for every non-empty CSR row (small DBs only)."""
    db = {}
    nz = np.nonzero(np.diff(offsets.astype(np.int64)))[0]
    for key in nz:
        b, e = int(offsets[key]), int(offsets[key + 1])
        db[int(key)] = [(int(v["branch"]), f32(v["score"])) for v in values[b:e]]
    return db
