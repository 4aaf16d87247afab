import json, sys
d = json.load(open(sys.argv[1]))
for w in ("wave0", "wave3"):
    print(w, d[w]["cycles_per_tile"])
    for k, v in d[w]["phases"].items():
        print("   %-66s %9.1f" % (k, v))
