# Runs on the GPU box (via gpurun): quick A/B of basic-version kernel variants, bench lines only.
# usage: bash tools/ab_basic.sh spec1 spec2 ...   spec = lib[@ENV=val[,ENV=val...]]; lib "main" = the product library,
# otherwise gstpeaq_amd/libpeaq_amd_<lib>.so (make -C gstpeaq_amd/csrc VARIANT=<lib> EXTRA=...).  Two interleaved passes.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for pass in 1 2; do
for spec in "$@"; do
  v=${spec%%@*}; envs=""; [ "$spec" != "$v" ] && envs=$(echo "${spec#*@}" | tr ',' ' ')
  tag=$(echo "$spec" | tr '@=,' '___')
  if [ "$v" = main ]; then lib=""; else lib="PEAQ_AMD_LIB=$R/gstpeaq_amd/libpeaq_amd_$v.so"; fi
  env $lib $envs timeout 300 python bench.py --steps ${AB_STEPS:-6} --warmup 2 --no-cpu-baseline ${AB_ARGS:---no-advanced} > $O/ab_${tag}_$pass.json 2> $O/ab_${tag}_$pass.err
  python - <<PY
import json
try:
    d = json.load(open("$O/ab_${tag}_$pass.json"))
    r = d["roofline"]
    print("%-40s pass $pass" % "$spec", "value %.3f M" % (d["value"] / 1e6), "ms/step %.2f" % d["ms_per_step"], "fe avg ms %.3f" % r["avg_launch_ms"],
          "be ms %.2f" % r.get("backend_ms", 0), ("adv %.3f M" % (d["advanced"]["value"] / 1e6)) if "advanced" in d else "")
except Exception as e:
    print("$spec FAILED", e, open("$O/ab_${tag}_$pass.err").read()[-600:])
PY
done
done
