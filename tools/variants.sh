# Runs on the GPU box (via gpurun): A/B of kernel variants built with `make -C gstpeaq_amd/csrc VARIANT=name`.
# usage: bash tools/variants.sh name1 name2 ...   ("main" = the product library)
# For every variant: the stage/golden parity tests, then the basic and advanced bench lines.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in "$@"; do
  if [ "$v" = main ]; then unset PEAQ_AMD_LIB; else export PEAQ_AMD_LIB=$R/gstpeaq_amd/libpeaq_amd_$v.so; fi
  echo "=== $v"
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline ${VARIANT_BENCH_ARGS:---no-advanced} > $O/var_$v.json 2> $O/var_$v.err
  python - <<PY
import json
try:
    d = json.load(open("$O/var_$v.json"))
    r = d["roofline"]
    print("$v", "value %.3f M" % (d["value"] / 1e6), "ms/step %.2f" % d["ms_per_step"], "fe avg ms %.3f" % r["avg_launch_ms"],
          "be ms %.2f" % r["backend_ms"], ("adv %.3f M" % (d["advanced"]["value"] / 1e6)) if "advanced" in d else "")
except Exception as e:
    print("$v FAILED", e, open("$O/var_$v.err").read()[-600:])
PY
done
