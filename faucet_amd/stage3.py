"""A consumer of the batched Stage-3 probes: JunctionMap::findNeighbor for MANY junctions in lock-step.

The reference's contig-graph stage (out of scope of this repository) walks the Bloom filter from every junction along every
extension a contig is built on: `findNeighbor` (utils/JunctionMap.cpp:231-412) asks `getValidJExtension` (:474-490) for the one
valid next base, advances, looks the k-mer up in the junction map, and repeats until it reaches a junction or a sink.  Each call is a
chain of dependent filter probes; different calls are independent.  Here all walks advance together: ONE call of
`fgpu_probe_valid_extension` per step answers `getValidJExtension` for every walk that is still under way (the device does the
4 x (contains + jcheck) probes of each), the host does what the reference does between two probes -- advance the DoubleKmer, look
the junction map up, keep the distance and contig-length bookkeeping.  No filter bit is tested on the host.

Restated from the reference's control flow (not its text); checked against the reference's own findNeighbor on golden maps
(oracle/_ref/ref_kat neighbors -> tests/golden/stage3_neighbors_*.jsonl.gz, tests/test_gpu_parity.py).
"""
from __future__ import annotations

import numpy as np


class NeighborWalker:
    def __init__(self, ctx, keys, recs, k: int, max_read_length: int):
        """keys / recs: the junction map (oriented k-mers; records with 'dist'), e.g. from Context.junctions() or a parsed .junctions"""
        self.ctx, self.k, self.mrl = ctx, k, max_read_length
        order = np.argsort(np.asarray(keys, dtype=np.uint64), kind="stable")
        self.keys = np.asarray(keys, dtype=np.uint64)[order]
        self.dist = np.asarray(recs["dist"], dtype=np.int64)[order]
        self.mask = np.uint64((1 << (2 * k)) - 1)
        self.steps = 0          # device calls made by the last find_neighbors
        self.probes = 0         # k-mers handed to getValidJExtension in them

    # ---- the reference's k-mer helpers on arrays (utils/Kmer.cpp:238-252,410-425; utils/DoubleKmer.cpp) ----
    def _revcomp(self, x):
        r = np.zeros_like(x)
        y = x.copy()
        for _ in range(self.k):
            r = (r << np.uint64(2)) | ((y & np.uint64(3)) ^ np.uint64(2))
            y >>= np.uint64(2)
        return r

    def _forward(self, km, rc, nuc):
        nuc = nuc.astype(np.uint64)
        return ((km << np.uint64(2)) | nuc) & self.mask, (rc >> np.uint64(2)) | ((nuc ^ np.uint64(2)) << np.uint64(2 * self.k - 2))

    def _is_junction(self, km):
        i = np.searchsorted(self.keys, km)
        i = np.minimum(i, len(self.keys) - 1)
        return self.keys[i] == km, i

    def find_neighbors(self, start_kmers, indices):
        """findNeighbor(junctionMap[start], start, index) for every pair; returns a structured array (kmer, node, rindex, dist, len,
        abort) in input order.  `abort` marks the calls in which the reference trips one of its asserts."""
        k = self.k
        n = len(start_kmers)
        km = np.asarray(start_kmers, dtype=np.uint64).copy()
        idx = np.asarray(indices, dtype=np.int64)
        rc = self._revcomp(km)
        found, where = self._is_junction(km)
        assert found.all(), "every start k-mer must be a junction of the map"
        maxd = self.dist[where, idx]
        out = np.zeros(n, dtype=[("kmer", np.uint64), ("node", np.int8), ("rindex", np.int8), ("dist", np.int32), ("len", np.int32), ("abort", np.int8)])
        done = np.zeros(n, dtype=bool)
        dist = np.ones(n, dtype=np.int64)
        ln = np.zeros(n, dtype=np.int64)
        lastnuc = np.zeros(n, dtype=np.int64)
        retidx = np.full(n, 4, dtype=np.int64)
        phase = np.zeros(n, dtype=np.int8)                      # 0: up to maxDist, 1: past it (overlapping sinks), 2: finished
        sink = np.zeros(n, dtype=out.dtype)

        def finish(m, kmer, node, rindex, d, length):
            out["kmer"][m], out["node"][m], out["rindex"][m], out["dist"][m], out["len"][m] = kmer, node, rindex, d, length
            done[m] = True
            phase[m] = 2

        # ---- the first one or two k-mers (:251-302)
        back = idx == 4
        km[back], rc[back] = rc[back].copy(), km[back].copy()                    # doubleKmer.reverse()
        ln[back] = k
        isj, _ = self._is_junction(km)
        m = back & isj
        finish(m, km[m], 1, 4, 1, ln[m])
        fw = ~back
        lastnuc[fw] = (rc[fw] & np.uint64(3)).astype(np.int64)
        nk, nr = self._forward(km[fw], rc[fw], idx[fw])
        km[fw], rc[fw] = nk, nr
        ln[fw] = 1 + k
        isj, _ = self._is_junction(rc)
        m = fw & isj
        finish(m, rc[m], 1, lastnuc[m], 1, ln[m])
        m = fw & ~done & (maxd == 1)
        finish(m, rc[m], 0, lastnuc[m], 1, ln[m])
        m = fw & ~done
        dist[m] = 2
        isj, _ = self._is_junction(km)
        m2 = m & isj
        finish(m2, km[m2], 1, 4, 2, ln[m2])
        m = ~done & (dist > maxd)                                                # the reference's assert(dist <= maxDist)
        out["abort"][m] = 1
        done[m] = True
        phase[m] = 2

        self.steps = self.probes = 0
        while True:
            # ---- leaving the first loop (:343-366): a junction exactly where expected, else this is (probably) a sink
            m = (phase == 0) & (dist >= maxd)
            if m.any():
                isj, _ = self._is_junction(km)
                j = m & isj
                finish(j, km[j], 1, retidx[j], dist[j], ln[j])
                s = m & ~isj
                sink["kmer"][s], sink["node"][s], sink["rindex"][s], sink["dist"][s], sink["len"][s] = km[s], 0, 4, dist[s], ln[s]
                phase[s] = 1
            m = (phase == 1) & (dist >= maxd + 2 * self.mrl)                     # :376, ran past every possible overlap
            out[m] = sink[m]
            phase[m] = 2
            act = np.nonzero(phase < 2)[0]
            if len(act) == 0:
                break
            # ---- ONE device call: getValidJExtension of every walk under way
            ext = self.ctx.probe_valid_extension(km[act]).astype(np.int64)
            self.steps += 1
            self.probes += len(act)
            bad = ext < 0
            a0 = act[bad & (phase[act] == 0)]                                    # assert(validExtension != -1 / -2)
            out["abort"][a0] = 1
            phase[a0] = 2
            b1 = act[bad & (phase[act] == 1)]                                    # off the real sequence: the sink stands
            out[b1] = sink[b1]
            phase[b1] = 2
            act, ext = act[~bad], ext[~bad]
            lastnuc[act] = (rc[act] & np.uint64(3)).astype(np.int64)
            nk, nr = self._forward(km[act], rc[act], ext)
            km[act], rc[act] = nk, nr
            ln[act] += 1
            # backward-facing half-step
            dist[act] += 1
            km[act], rc[act] = rc[act].copy(), km[act].copy()
            retidx[act] = lastnuc[act]
            isj, wj = self._is_junction(km[act])
            a = phase[act] == 0
            stop = a & (dist[act] == maxd[act])                                  # break: handled at the top of the next round
            hit = a & ~stop & isj
            finish(act[hit], km[act[hit]], 1, retidx[act[hit]], dist[act[hit]], ln[act[hit]])
            hb = ~a & isj                                                        # past maxDist: overlap test (:391-404)
            if hb.any():
                w = act[hb]
                overlap = self.dist[wj[hb], lastnuc[w]] + maxd[w] - dist[w]
                ok = overlap >= 0
                finish(w[ok], km[w[ok]], 1, lastnuc[w[ok]], dist[w[ok]], ln[w[ok]])
                out[w[~ok]] = sink[w[~ok]]
                phase[w[~ok]] = 2
            go = (phase[act] < 2) & ~stop
            act = act[go]
            # forward-facing half-step
            dist[act] += 1
            km[act], rc[act] = rc[act].copy(), km[act].copy()
            retidx[act] = 4
            isj, wj = self._is_junction(km[act])
            a = phase[act] == 0
            stop = a & (dist[act] == maxd[act])
            hit = a & ~stop & isj
            finish(act[hit], km[act[hit]], 1, 4, dist[act[hit]], ln[act[hit]])
            hb = ~a & isj
            if hb.any():
                w = act[hb]
                overlap = self.dist[wj[hb], 4] + maxd[w] - dist[w]
                ok = overlap >= 0
                finish(w[ok], km[w[ok]], 1, 4, dist[w[ok]], ln[w[ok]])
                out[w[~ok]] = sink[w[~ok]]
                phase[w[~ok]] = 2
        return out
