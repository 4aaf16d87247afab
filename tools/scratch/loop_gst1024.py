# development: the 1024-element gst-launch process of tests/test_gpu_broker_1024.py in a loop, to catch a rare hang
import re, subprocess, sys, time
sys.path.insert(0, "tests")
import gst_env
n = 1024
args = []
for i in range(n):
    waves = ("sine", "sine") if i % 2 == 0 else ("saw", "triangle")
    args += ["audiotestsrc", f"name=s{i}", "num-buffers=128", f"wave={waves[0]}", "freq=440",
             "audiotestsrc", f"name=r{i}", "num-buffers=128", f"wave={waves[1]}", "freq=440",
             "peaq", f"name=p{i}", f"s{i}.src!p{i}.ref", f"r{i}.src!p{i}.test"]
env = gst_env.env(); env["PEAQ_AMD_BROKER"] = str(n)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for k in range(reps):
    t0 = time.time()
    try:
        out = subprocess.run(["gst-launch-1.0", "-q", f"--gst-plugin-load={gst_env.PLUGIN}", *args], capture_output=True, text=True, env=env, timeout=90)
        odgs = re.findall(r"Objective Difference Grade: (-?[0-9.]+|-?nan)", out.stdout)
        ok = out.returncode == 0 and sorted(odgs) == sorted(["0.171"] * (n // 2) + ["-2.007"] * (n // 2))
        print(k, "ok" if ok else "BAD rc=%d n=%d %s" % (out.returncode, len(odgs), out.stderr[-300:]), round(time.time() - t0, 1), flush=True)
    except subprocess.TimeoutExpired as e:
        print(k, "HANG after 90 s; stderr tail:", (e.stderr or b"")[-500:], flush=True)
