import json,sys
for l in open(sys.argv[1]):
    try: d=json.loads(l)
    except Exception: continue
    print(d["config"]["workload"][:24], "| %8.0f Mrays/s | %7.2f ms | integrate %7.2f ms | vgpr %s lds %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_hip_events"], d["config"]["vgprs"], d["config"]["lds_bytes_per_wave"]), d["config"].get("parity_full_frame",""))
