#!/bin/bash
# ON THE GPU BOX: the tree against a second build of an earlier commit kept under ab_old/ (git worktree, not tracked), same box, alternating.
#   in the build container first:  git worktree add -f ab_old <commit> && echo ab_old/ >> .git/info/exclude && (cd ab_old && python -m faucet_amd.build && make -C oracle)
#   afterwards:                    git worktree remove --force ab_old
#   full-size legs included: configs 2, 5, 4 and the CLI legs from one bench run each
root=${GRAFT_REPO_ROOT:-$(pwd)}
show() {
python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
o=[sys.argv[2], 'step %.2f ms' % d['ms_per_step']]
if 'cli_file_to_files' in d: o.append('cli %.3f s' % d['cli_file_to_files']['seconds'])
if 'config3_cli' in d and 'pass_ms' in d['config3_cli']: o.append('config3 passes %s' % list(d['config3_cli']['pass_ms'].values()))
for n in ('config5','config4'):
    f=d.get('full_size',{}).get(n)
    if f and 'seconds' in f: o.append('%s %.3f s (walk_stage %s, load_mark %s)' % (n, f['seconds'], f['second_step']['kernel_ms'].get('walk_stage'), f['second_step']['kernel_ms'].get('load_mark')))
print(' | '.join(o))
" "$1" "$2"
}
for i in 1 2; do
  for t in new old; do
    d=$root; [ $t = old ] && d=$root/ab_old
    (cd $d && python3 bench.py --no-cpu > $root/gpurun_out/ab_$t$i.json 2> /dev/null)
    show $root/gpurun_out/ab_$t$i.json "$t run $i"
  done
done
