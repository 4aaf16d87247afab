# GPU box: A/B of the advanced pass (default engine) between the product library and variant libraries, same box
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do
  for v in "" $*; do
    lib=$R/gstpeaq_amd/libpeaq_amd${v:+_$v}.so
    PEAQ_AMD_LIB=$lib python bench.py --advanced --no-scaling-reference --no-cpu-baseline --steps 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('adv ${v:-main}', round(d['value']/1e6,3), round(d['ms_per_step'],2), 'bank', round(d['roofline']['avg_launch_ms'],1), 'nan', d.get('odg_nan'), 'odg', d.get('odg_mean'))"
  done
done
