"""The multi-GPU host protocol (faucet_amd/sharded.py) on CPU: 2 processes, gloo, with an oracle-backed stand-in for the
device backend.  Checks that the exchange steps (presence bitmaps -> exclusive prefix-OR -> ordered load -> OR-allreduce;
pure scan everywhere -> walk handed rank to rank) reproduce the single-process result byte for byte."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from faucet_amd import _lib as L
from faucet_amd import sharded
from oracle import pyoracle as po
from tests.golden_util import Case

ENTRY = np.dtype([("key", np.uint64), ("stamp", np.uint64), ("dist", np.uint8, 5), ("cov", np.uint8, 4), ("linked", np.uint8),
                  ("pad", np.uint8, 6)])
assert ENTRY.itemsize == L.TABLE_ENTRY_BYTES


class OracleShard:
    """CPU stand-in with the method set of sharded.GpuShard (TEST ONLY: the product backend is GpuShard)."""

    def __init__(self, k, tai, nh, j, spacer, protocol="fixup"):
        self.k, self.tai, self.nh, self.j, self.spacer, self.protocol = k, tai, nh, j, spacer, protocol
        self.b1, self.b2 = po.Bloom(tai, nh), po.Bloom(tai, nh)
        self.sc = None
        self.prepared = []

    def fence(self):
        pass

    def scratch(self, nbytes, tag="gather"):
        return torch.empty(nbytes, dtype=torch.uint8)

    def header_tensor(self):
        return torch.zeros(sharded.HEADER_WORDS, dtype=torch.int64)

    def clear_filters(self):
        self.b1.bits()[:] = 0
        self.b2.bits()[:] = 0

    def presence(self, batch):
        po.load_single_filter(self.b1, batch[0], batch[1], self.k)

    def bloom_tensor(self, which):
        return torch.from_numpy((self.b1 if which == L.BLOO1 else self.b2).bits())

    def or_into(self, which, src):
        (self.b1 if which == L.BLOO1 else self.b2).bits()[:] |= src.numpy()

    def load(self, batches, keep_carry, shard_times=False):
        if not keep_carry:
            self.clear_filters()
        self._batches = batches
        st = None
        for b in batches:
            st = po.load_two_filters(self.b1, self.b2, b[0], b[1], self.k)
        return {"kmers": int(st.kmers)}

    def fixup_possible(self, batches):
        self.asked = True
        return True

    def fixup_ready(self):
        # protocol "fallback": the last rank's own load "did not stay resident" -- the ranks vote and ALL take the presence protocol
        return not (self.protocol == "fallback" and dist.get_rank() == dist.get_world_size() - 1)

    def or_tensor(self, dst, src):
        dst.numpy()[:] |= src.numpy()

    def load_fixup(self, prefix):
        # what the device's fix-up must amount to: the shard loaded with the prefix as the carried-in state (the device gets there
        # from its own first-set times; the parity of THAT kernel with this is tests/test_gpu_parity.py's business)
        self.fixed_up = True
        local = self.b1.bits().copy()
        self.b1.bits()[:] = prefix.numpy()
        self.b2.bits()[:] = 0
        st = None
        for b in self._batches:
            st = po.load_two_filters(self.b1, self.b2, b[0], b[1], self.k)
        assert np.array_equal(self.b1.bits(), local | prefix.numpy())
        return {"kmers": int(st.kmers)}

    def scan_begin(self):
        self.sc = po.Scanner(self.k, self.j, self.spacer, self.b2)
        self.prepared = []

    def scan_prepare(self, batch):
        self.prepared.append(batch)

    def scan_stream(self, batches, after_batch=None):
        for i, batch in enumerate(batches):
            self.sc.scan_reads(batch[0], batch[1])
            if after_batch:
                after_batch(i)
        return self.scan_end()

    def import_hint(self, buf, n):
        # the oracle has no preview to feed; what matters here is that the hint is a state of rank 0's table and never part of a result
        self.hint = buf.numpy()[: n * L.TABLE_ENTRY_BYTES].view(ENTRY).copy()

    def refresh_prepared(self):
        # (round 5: a rank passes the table it is handed on as a fresher preview; the oracle prepares nothing ahead, the check below is the point)
        self.late_hints = getattr(self, "late_hints", 0) + 1

    def walk_shard(self, batches, buf, n, carried):
        self.import_table(buf, n, carried)
        if getattr(self, "hint", None) is not None and len(self.hint):   # the hint is an EARLIER state of the first shard's table
            real = buf.numpy()[: n * L.TABLE_ENTRY_BYTES].view(ENTRY)
            pos = {int(k): i for i, k in enumerate(real["key"])}
            for e in self.hint:
                r = real[pos[int(e["key"])]]                              # every key is still there ...
                assert (e["dist"] <= r["dist"]).all()                     # ... and distances have only grown
        self.scan_walk_prepared()
        return self.scan_end()

    def scan_walk_prepared(self):
        for b in self.prepared:
            self.sc.scan_reads(b[0], b[1])

    def scan_end(self):
        st = self.sc.stats()
        return {n: st.get(n, 0) for n in sharded._STAT_NAMES}

    def import_table(self, buf, n, carried):
        e = buf.numpy()[: n * L.TABLE_ENTRY_BYTES].view(ENTRY)
        recs = np.zeros(n, dtype=po.JUNC_DTYPE)
        recs["dist"], recs["cov"] = e["dist"], e["cov"]
        recs["linked"] = (e["linked"][:, None] >> np.arange(5)) & 1
        order = np.argsort(e["stamp"], kind="stable")
        oc = {k: carried[k] for k in ("reads_processed", "unambiguous_reads", "reads_no_errors", "nb_jcheck_kmer", "nb_no_juncs",
                                      "nb_processed", "nb_skipped")}
        oc.update(empty_count=0, not_empty_count=0, n_junctions=0)
        self.sc.import_junctions(e["key"][order], recs[order], oc)

    def export_table(self, tag="table_out"):
        keys, recs = self.sc.junctions("creation")
        e = np.zeros(max(len(keys), 1), dtype=ENTRY)
        e["key"][: len(keys)] = keys
        e["stamp"][: len(keys)] = np.arange(len(keys))
        e["dist"][: len(keys)], e["cov"][: len(keys)] = recs["dist"], recs["cov"]
        e["linked"][: len(keys)] = (recs["linked"].astype(np.uint8) << np.arange(5, dtype=np.uint8)).sum(axis=1)
        return len(keys), torch.from_numpy(e.view(np.uint8).reshape(-1).copy())

    def junctions(self):
        return self.sc.junctions("creation")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, out_dir, protocol):
    os.environ["FAUCET_SHARD_PROTOCOL"] = "auto" if protocol == "fallback" else protocol
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # one node: no host-name look-up for the interface
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = Case(name)
    bases, offs = po.reads_from_lines(c.lines())
    n = len(offs) - 1
    cuts = np.linspace(0, n, world + 1).astype(int)
    lo, hi = cuts[rank], cuts[rank + 1]
    mine = [(bases, offs[lo:(lo + hi) // 2 + 1].copy()), (bases, offs[(lo + hi) // 2:hi + 1].copy())]   # two batches per shard
    tai, nh, _, _ = po.sizing_from_cli(c.E, c.S)
    be = OracleShard(c.k, tai, nh, c.j, c.spacer, protocol)
    sharded.load_sharded(be, mine, rank, world)
    assert getattr(be, "fixed_up", False) == (protocol not in ("presence", "fallback") and rank > 0)      # auto: the fix-up protocol wherever every rank can run it
    np.save(os.path.join(out_dir, f"bloo2_{rank}.npy"), be.b2.bits().copy())
    st, last = sharded.scan_sharded(be, mine, rank, world)
    if last:
        keys, recs = be.junctions()
        lines = po.junction_lines(keys, recs, c.k)
        with open(os.path.join(out_dir, "junctions.txt"), "w") as f:
            f.write("\n".join(lines))
        np.save(os.path.join(out_dir, "stats.npy"), np.array([st["nb_processed"], st["nb_skipped"], st["nb_jcheck_kmer"], st["nb_no_juncs"],
                                                               st["reads_no_errors"], st["reads_processed"]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("protocol", ["fixup", "presence", "auto", "fallback"])
@pytest.mark.parametrize("name,world", [("c1_k21", 2), ("ragged_k31", 2), ("j2_spacer20_k15", 3)])
def test_sharded_protocol_matches_single_process(name, world, protocol, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path), protocol), nprocs=world, join=True)
    c = Case(name)
    for r in range(world):      # every rank ends up with the reference's bloo2
        assert np.array_equal(np.load(tmp_path / f"bloo2_{r}.npy"), c.bloom())
    got = (tmp_path / "junctions.txt").read_text().split("\n")
    assert sorted(got) == sorted(c.junction_lines())
    cn = c.counters
    st = np.load(tmp_path / "stats.npy")
    assert list(st) == [cn["nb_processed"], cn["nb_skipped"], cn["nb_jcheck_kmer"], cn["nb_no_juncs"], cn["reads_no_errors"],
                        cn["scan_reads_processed"]]


class _CpuOr:
    """the two backend methods the bitmap exchanges use, on CPU tensors"""

    def fence(self):
        pass

    def scratch(self, nbytes, tag="gather"):
        return torch.full((nbytes,), 0xA5, dtype=torch.uint8)       # garbage on purpose: the exchanges must initialise what they read

    def or_tensor(self, dst, src):
        dst.numpy()[:] |= src.numpy()


def _exchange_worker(rank, world, port, nbytes, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # one node: no host-name look-up for the interface
    dist.init_process_group("gloo", rank=rank, world_size=world)
    maps = [np.random.default_rng(100 + r).integers(0, 256, size=nbytes, dtype=np.uint8) & np.random.default_rng(200 + r).integers(0, 256, size=nbytes, dtype=np.uint8)
            for r in range(world)]
    be = _CpuOr()
    mine = torch.from_numpy(maps[rank].copy())
    prefix = torch.full((nbytes,), 0x5A, dtype=torch.uint8)
    sharded.exclusive_prefix_or(be, mine, prefix, rank, world)
    want = np.zeros(nbytes, dtype=np.uint8)
    for r in range(rank):
        want |= maps[r]
    ok = np.array_equal(prefix.numpy(), want) and np.array_equal(mine.numpy(), maps[rank])      # the input is left alone
    sharded.or_allreduce(be, mine, rank, world)
    total = np.zeros(nbytes, dtype=np.uint8)
    for r in range(world):
        total |= maps[r]
    ok = ok and np.array_equal(mine.numpy(), total)
    with open(os.path.join(out_dir, f"ok_{rank}"), "w") as f:
        f.write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nbytes", [(2, 4096), (3, 4096), (3, 64), (4, 16), (3, 1 << 20)])
def test_bitmap_exchanges_by_slices(world, nbytes, tmp_path):
    """exclusive prefix-OR and OR-allreduce of the filters' bit arrays as sharded.py runs them: slices collected per rank (grouped
    send/recv), reduced locally, handed back / gathered -- including slice counts that do not divide the bitmap and ranks whose slice is empty"""
    mp.spawn(_exchange_worker, args=(world, _free_port(), nbytes, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok_{r}").read_text() == "1" for r in range(world))


@pytest.mark.parametrize("protocol", ["presence", "fixup"])
@pytest.mark.parametrize("name,world", [("c1_k21", 3), ("ragged_k31", 2), ("j2_spacer20_k15", 4)])
def test_ranks_in_turn_in_one_process_match_the_sequential_run(name, world, protocol):
    """sharded.run_in_turn (how tests/test_gpu_fullsize.py runs BASELINE config 4's eight shards on one GPU): after every rank the filters
    and the junction map are the SEQUENTIAL run's after that shard -- checked here against the oracle run over the same prefix of the reads"""
    c = Case(name)
    bases, offs = po.reads_from_lines(c.lines())
    n = len(offs) - 1
    cuts = np.linspace(0, n, world + 1).astype(int)
    shards = []
    for r in range(world):
        lo, hi = cuts[r], cuts[r + 1]
        shards.append([(bases, offs[lo:(lo + hi) // 2 + 1].copy()), (bases, offs[(lo + hi) // 2:hi + 1].copy())])
    tai, nh, _, _ = po.sizing_from_cli(c.E, c.S)
    s1, s2 = po.Bloom(tai, nh), po.Bloom(tai, nh)                    # the sequential run, advanced shard by shard
    seen = {"load": 0, "scan": 0}

    def after_load(r, stats, bloo1, bloo2):
        po.load_two_filters(s1, s2, bases, offs[cuts[r]:cuts[r + 1] + 1].copy(), c.k)
        assert np.array_equal(bloo1.numpy(), s1.bits()) and np.array_equal(bloo2.numpy(), s2.bits()), f"filters after shard {r}"
        seen["load"] += 1

    seq = {}

    def after_scan(r, stats, backend):
        if "sc" not in seq:
            seq["sc"] = po.Scanner(c.k, c.j, c.spacer, s2)           # s2 is final by now: all loads come before the first scan
        seq["sc"].scan_reads(bases, offs[cuts[r]:cuts[r + 1] + 1].copy())
        keys, recs = backend.junctions()
        okeys, orecs = seq["sc"].junctions("creation")
        assert np.array_equal(keys, okeys) and np.array_equal(recs, orecs), f"junction map after shard {r}"
        ost = seq["sc"].stats()
        for key in ("nb_processed", "nb_skipped", "nb_jcheck_kmer", "nb_no_juncs", "reads_no_errors", "reads_processed"):
            assert stats[key] == ost[key], (r, key)
        seen["scan"] += 1

    lst, st, last = sharded.run_in_turn(lambda: OracleShard(c.k, tai, nh, c.j, c.spacer, protocol), shards, protocol, after_load, after_scan)
    assert seen == {"load": world, "scan": world}
    assert np.array_equal(s2.bits(), c.bloom())
    assert sorted(po.junction_lines(*last.junctions(), c.k)) == sorted(c.junction_lines())
