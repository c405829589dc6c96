import json,sys
for f in sys.argv[1:]:
    try:
        l=[x for x in open(f) if x.startswith("{")][-1]; d=json.loads(l)
    except Exception as e:
        print(f, "ERR", e); continue
    print(f.split("/")[-1], d["config"]["shortlist"], d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], d["config"]["rescued_queries"])
    for k,v in d.get("other_shortlists",{}).items(): print("   ",k, v["value"], v["ms_per_step"], v["roofline"]["launch_ms"], v["fused_top10_identical_to_primary"], v["rescued_queries"])
