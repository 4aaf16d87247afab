# GPU box: A/B of the basic step between the product library and variant libraries, alternating, same box
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in 1 2; do
  for v in "" $*; do
    lib=$R/gstpeaq_amd/libpeaq_amd${v:+_$v}.so
    PEAQ_AMD_LIB=$lib python bench.py --no-advanced --no-scaling-reference --no-cpu-baseline --steps 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('basic ${v:-main}', round(d['value']/1e6,3), round(d['ms_per_step'],2))"
  done
done
