# tools/trace_dens.sh [-w warmup] lib... : mean duration of the step's kernels over 1000 steps of cfg2 after `warmup` (default 200) steps,
# per build (rocprofv3 kernel trace; run on the GPU box)
W=200; if [ "$1" = "-w" ]; then W=$2; shift 2; fi
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/tr_$lib
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$lib -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-also --steps 1000 --warmup $W --lib $GRAFT_REPO_ROOT/pi-sph-fluid_amd/csrc/$lib > /tmp/tr_$lib.log 2>&1
  python3 - $lib $W <<'PY'
import csv, statistics, sys
lib, W = sys.argv[1], int(sys.argv[2])
tr=list(csv.DictReader(open('/tmp/tr_%s/trace_kernel_trace.csv'%lib)))
tr.sort(key=lambda r:int(r['Start_Timestamp']))
out=[lib]
# the timed window: the 1000 launches of the force pass before the last 350 (the kernel timings that follow the window)
names=[('dens','k_density_list<1, 0'),('check','k_check'),('verify','k_verify'),('gate','k_rebuild<0>'),('force','k_force_list<2, 0>')]
f=[r for r in tr if 'k_force_list<2, 0>' in r['Kernel_Name']]
# window start/end timestamps from the step structure: the (W+1)th .. (W+1000)th force launch of the stepping phase
t0=int(f[W]['Start_Timestamp']) if len(f)>W+1000 else 0
t1=int(f[W+999]['End_Timestamp']) if len(f)>W+1000 else 1<<62
for k,nm in names:
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in tr if nm in r['Kernel_Name'] and t0<=int(r['Start_Timestamp'])<=t1]
    if d: out.append('%s n %d mean %.2f med %.2f p90 %.1f' % (k, len(d), statistics.mean(d), statistics.median(d), sorted(d)[int(len(d)*0.9)]))
out.append('step %.2f us' % ((t1-t0)/1e3/1000))
print(' | '.join(out))
PY
done
