# GPU box: phase profile of the FP64 bank kernel, a stage parity check and the advanced bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
TAG=${1:-x}
PEAQ_AMD_LIB=$R/gstpeaq_amd/libpeaq_amd_fbprof.so python tools/fb_profile.py 1024 f64 > gpurun_out/r04_fb_phases_$TAG.json 2>/dev/null
python tools/scratch/show_fb_phases.py gpurun_out/r04_fb_phases_$TAG.json | head -17
python tools/scratch/dbg_fb_stage.py 2>/dev/null | tail -40 | sort -k3 -g | tail -2
python bench.py --advanced --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-reference 2>/dev/null > gpurun_out/r04_adv_$TAG.json
python - <<PY
import json
d=json.load(open("gpurun_out/r04_adv_$TAG.json")); print("value %.3f M  ms/step %.1f bank %.1f ms" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["avg_launch_ms"]), "nan", d.get("odg_nan"))
PY
