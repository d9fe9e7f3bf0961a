for cs in "1024 1048576" "256 4194304" "4096 262144" "512 2097152" "1024 1048576"; do set -- $cs; timeout -k 10 200 python3 bench.py --no-cpu-baseline --channels $1 --samples $2 --steps 10 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 x $2', r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], r['roofline']['rest_of_call_ms'])" || exit 1; done
