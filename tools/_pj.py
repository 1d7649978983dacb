import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1], d["ms_per_step"], d["ms_per_step_repeated_batch"], d["config"]["critic_term_recordings"], d["roofline"] and d["roofline"]["frac"], d["sample_nodes_per_sec"])
