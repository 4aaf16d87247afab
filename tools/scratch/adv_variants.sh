R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for v in "$@"; do
  if [ "$v" = main ]; then unset PEAQ_AMD_LIB; else export PEAQ_AMD_LIB=$R/gstpeaq_amd/libpeaq_amd_$v.so; fi
  for mode in "" "--reduced-precision"; do
  python bench.py --advanced $mode --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-reference 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v $mode: value %.3f M  ms/step %.1f bank %.1f ms' % (d['value']/1e6, d['ms_per_step'], r['avg_launch_ms']), 'nan', d.get('odg_nan'))"
  done
done
