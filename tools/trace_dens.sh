# tools/trace_dens.sh lib... : mean duration of the step's three kernels over steps 200-1200 of cfg2, per build (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/tr_$lib
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$lib -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-also --steps 1000 --warmup 200 --lib $GRAFT_REPO_ROOT/pi-sph-fluid_amd/csrc/$lib > /tmp/tr_$lib.log 2>&1
  python3 - $lib <<'PY'
import csv, statistics, sys
lib=sys.argv[1]
tr=list(csv.DictReader(open('/tmp/tr_%s/trace_kernel_trace.csv'%lib)))
tr.sort(key=lambda r:int(r['Start_Timestamp']))
out=[lib]
for k,nm in (('dens','k_density_list<1, 0, true>'),('gate','k_rebuild<0>'),('force','k_force_list<2, 0>')):
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in tr if nm in r['Kernel_Name']]
    d=d[200:1200]
    out.append('%s mean %.2f med %.2f p90 %.1f' % (k, statistics.mean(d), statistics.median(d), sorted(d)[int(len(d)*0.9)]))
print(' | '.join(out))
PY
done
