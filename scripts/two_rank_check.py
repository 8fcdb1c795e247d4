"""Two RANKS, two processes, ONE GPU (diagnostic for boxes without a second GPU):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 scripts/two_rank_check.py [reads_per_rank]

Runs the read-sharded pipeline of faucet_amd/sharded.py with the product backend (GpuShard over libfaucet_gpu.so) in two
processes that share cuda:0, the exchange steps carried by gloo through host memory instead of RCCL, and compares the
outcome (bloo2 bytes, junction records in creation order, counters) with ONE context that is fed both shards in file
order.  Everything except the transport is what `bench.py --gpus 2` runs.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faucet_amd import _lib as L  # noqa: E402
from faucet_amd import api, sharded  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # one node: do not let gloo look the host name up to find an interface (it may not resolve)
dist.init_process_group("gloo")
dist.barrier()
print(f"INIT OK rank {rank}", flush=True)      # from here on a stall is the product's, not the rendezvous' (tests/test_gpu_fullsize.py tells them apart)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
k, Lr = 31, 100
tai, nh = api.load_filter_shape(10 * n * world, 2 * n * world)
genome = bench.make_genome(2 * n * world, 2, dev)
reads = bench.make_reads(genome, n, Lr, 0.01, 1000 + rank, dev)
batches = bench.device_batches(reads, 1_000_000)
PAIRS = os.environ.get("FAUCET_CHECK_PAIRS", "0") == "1"      # both pair filters on, handed from rank to rank with the table (consecutive reads = pairs)
ctx_kw = {"record_stops": True} if PAIRS else {}
if PAIRS:
    _, stai, snh = api.size_optimal(max(n * world // 20, 1000), np.float32(0.01))
    _, ltai, lnh = api.size_optimal(max(n * world // 10, 1000), np.float32(0.01))
if os.environ.get("FAUCET_TORCH_STREAM", "0") == "1":      # the library on torch's stream, no host fences (what bench.py does for N > 1)
    ts = torch.cuda.Stream(dev)
    torch.cuda.set_stream(ts)
    ctx = api.Context(k, tai, nh, device=0, stream=ts.cuda_stream, **ctx_kw)
else:
    ctx = api.Context(k, tai, nh, device=0, **ctx_kw)
shard = sharded.GpuShard(ctx, dev)
if PAIRS:
    shard.pairs_setup(short=(stai, snh), long=(ltai, lnh))
lst = sharded.load_sharded(shard, batches, rank, world)
bloo2 = ctx.bloom_download(L.BLOO2)
sst, last = sharded.scan_sharded(shard, batches, rank, world)
res = {"rank": rank, "to_bloo2": lst["to_bloo2"], "kmers": lst["kmers"]}
if PAIRS:
    res["pair_counts"] = shard.pair_counts()
if last:
    keys, recs = shard.junctions()
    # the same reads through one context, in file order (rank 0's shard, then rank 1's, ...)
    allreads = torch.cat([bench.make_reads(genome, n, Lr, 0.01, 1000 + r, dev) for r in range(world)])
    one = api.Context(k, tai, nh, device=0, **ctx_kw)
    if PAIRS:
        one.scan_short_pairs(stai, snh, lists_to_host=False)
        one.scan_long_pairs(ltai, lnh, 2)
    olst, osst, ob2, okeys, orecs = bench.step_single(one, bench.device_batches(allreads, 1_000_000))
    same_b2 = bool(np.array_equal(bloo2, ob2))
    same_keys = bool(np.array_equal(keys, okeys))
    same_recs = same_keys and all(np.array_equal(recs[f], orecs[f]) for f in ("dist", "cov", "linked"))
    counters = all(sst[c] == osst[c] for c in ("n_junctions", "nb_jcheck_kmer", "nb_no_juncs", "nb_processed", "nb_skipped",
                                                "reads_no_errors", "reads_processed", "unambiguous_reads"))
    print(f"two ranks vs one context: bloo2 {same_b2}, junction keys {same_keys} ({len(keys)} vs {len(okeys)}), records {same_recs}, "
          f"counters {counters}", flush=True)
    res["ok"] = same_b2 and same_keys and same_recs and counters
    if PAIRS:
        lbits, oe, one_ne = one.scan_long_pairs_download(ltai)
        mine_long, _, _ = ctx.scan_long_pairs_download(ltai)
        same_short = bool(np.array_equal(ctx.scan_short_pairs_download(stai), one.scan_short_pairs_download(stai)))
        same_long = bool(np.array_equal(mine_long, lbits)) and bool(lbits.any())
        res["one_counts"] = (oe, one_ne)
        print(f"pair filters: short {same_short}, long {same_long}", flush=True)
        res["ok"] = res["ok"] and same_short and same_long
out = [None] * world
dist.all_gather_object(out, res)
if rank == 0:
    total = sum(o["to_bloo2"] for o in out)
    ok = [o.get("ok") for o in out if "ok" in o][0]
    if PAIRS:
        want = [o["one_counts"] for o in out if "one_counts" in o][0]
        got = (sum(o["pair_counts"][0] for o in out), sum(o["pair_counts"][1] for o in out))
        print("pair counts of the ranks", got, "of one context", tuple(want), flush=True)
        ok = ok and got == tuple(want)
    print("to_bloo2 per rank", [o["to_bloo2"] for o in out], "sum", total, "RESULT", "PASS" if ok else "FAIL", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0)
