import json, sys
d = json.load(sys.stdin)
print(sys.argv[1] if len(sys.argv) > 1 else "", "fps", d["value"], "ms", d["ms_per_step"], "eager_ms", d["config"].get("eager_ms_per_step_with_event_timing"), "dom", d["roofline"]["kernel"], d["roofline"]["achieved"])
