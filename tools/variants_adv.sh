# Runs on the GPU box (via gpurun): A/B of filter-bank kernel variants (advanced version), see tools/variants.sh.
# usage: bash tools/variants_adv.sh name1 name2 ...   ("main" = the product library)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in "$@"; do
  if [ "$v" = main ]; then unset PEAQ_AMD_LIB; else export PEAQ_AMD_LIB=$R/gstpeaq_amd/libpeaq_amd_$v.so; fi
  echo "=== $v"
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fir_modes.py -x -q 2>&1 | tail -3
  timeout 300 python bench.py ${VARIANT_ADV_ARGS:-} --advanced --steps 3 --warmup 1 --no-cpu-baseline > $O/vara_$v.json 2> $O/vara_$v.err
  python - <<PY
import json
try:
    d = json.load(open("$O/vara_$v.json"))
    r = d["roofline"]
    print("$v", "value %.3f M" % (d["value"] / 1e6), "ms/step %.2f" % d["ms_per_step"], {k: r[k] for k in r if k.endswith("_ms")})
except Exception as e:
    print("$v FAILED", e, open("$O/vara_$v.err").read()[-600:])
PY
done
