"""Pass 1 alone with the two layouts of its working state (load.hip, Filt: the interleaved pair + first[] / 256-byte records), kernel times and the
filters' digests (they must agree): BASELINE config 4's per-GPU shape (25 M reads of a 400 Mb genome, 2 x 1 GiB filters), config 5's first 10 M reads,
config 2.  GPU box.  usage: python scripts/load_layout_ab.py [shape ...]   (shapes: c4 c5 c2; default all)
The layout is chosen per process (FGPU_LOAD_LAYOUT is read once): every measurement is a child process, the two layouts alternating."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = {
    "c4": dict(genome=400_000_000, gseed=4, reads=25_000_000, read_len=100, err=0.01, rseed=4000, E=1_000_000_000, S=200_000_000, batch=2_500_000),
    "c5": dict(genome=150_000_000, gseed=5, reads=10_000_000, read_len=150, err=0.05, rseed=5000, E=2_000_000_000, S=1_000_000_000, batch=2_000_000),
    "c2": dict(genome=20_000_000, gseed=2, reads=10_000_000, read_len=100, err=0.01, rseed=1000, E=100_000_000, S=20_000_000, batch=1_000_000),
}


def child(shape):
    import torch
    import bench
    from faucet_amd import _lib as L
    from faucet_amd import api
    from faucet_amd import synth_det as sd
    c = SHAPES[shape]
    dev = torch.device("cuda", 0)
    g = sd.make_genome(c["genome"], c["gseed"], dev)
    reads = sd.make_reads(g, c["reads"], c["read_len"], c["err"], c["rseed"], dev)
    del g
    tai, nh = api.load_filter_shape(c["E"], c["S"])
    batches = bench.device_batches(reads, bench.batch_bounds(c["reads"], c["batch"], 2))
    ctx = api.Context(31, tai, nh, profile=True)
    for rep in range(3):
        ctx.kernel_times_reset()
        ctx.load_begin()
        for b in batches:
            ctx.load_batch(b)
        st = ctx.load_end()
    t = ctx.kernel_times()
    ms = lambda k: t.get(k, (0, 0.0))[1]
    d1 = hashlib.sha256(ctx.bloom_download(L.BLOO1).tobytes()).hexdigest()[:16]
    d2 = hashlib.sha256(ctx.bloom_download(L.BLOO2).tobytes()).hexdigest()[:16]
    tag = os.environ.get('FGPU_LOAD_LAYOUT', 'default') + ''.join(f" {k[5:].lower()}={os.environ[k]}" for k in ("FGPU_SWEEP_RATIO",) if k in os.environ)
    print(f"{shape} {tag:34s} tai 2^{tai.bit_length() - 1} n_hash {nh}: load_mark {ms('load_mark'):8.2f}  load_resolve {ms('load_resolve'):7.2f}  "
          f"carry_update {ms('carry_update'):7.2f}  pair_join {ms('pair_join'):6.2f}  pair_split {ms('pair_split'):6.2f}  sum {ms('load_mark') + ms('load_resolve') + ms('carry_update') + ms('pair_join') + ms('pair_split'):8.2f} ms per pass; "
          f"to_bloo2 {st['to_bloo2']} bloo1 {d1} bloo2 {d2}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        variants = [dict(FGPU_LOAD_LAYOUT="pair"), dict(FGPU_LOAD_LAYOUT="records"), dict(FGPU_LOAD_LAYOUT="pair"), dict(FGPU_LOAD_LAYOUT="records"),
                    dict(FGPU_LOAD_LAYOUT="records", FGPU_SWEEP_RATIO="1/1")]
        for shape in (sys.argv[1:] or ["c4", "c5", "c2"]):
            for v in variants:
                subprocess.run([sys.executable, os.path.abspath(__file__), "--child", shape], env=dict(os.environ, **v), cwd=ROOT)
