run() { echo "== $*"; env "$@" python bench.py --steps 2 --no-cpu-baseline --no-pair-merge --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read());print(d['value'],d['ms_per_step'],d['phase_seconds_last_build'])"; }
run KSH_LANES=3
run KSH_LANES=4
run KSH_LANES=5
run KSH_LANES=6
run KSH_LANES=8
run KSH_LANES=4 KSH_LANE_CUS=2
run KSH_LANES=5 KSH_LANE_CUS=2
run KSH_LANES=3
