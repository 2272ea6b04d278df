for round in 1 2; do
for WL in c2 c3; do for F in 1 2 4 8 16 32; do echo -n "wl=$WL frames=$F : "; timeout -k 10 120 python3 bench.py --steps 60 --no-cpu-baseline --workload $WL --frames $F 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['roofline']['frac'], 'us', r['roofline']['avg_launch_us'], 'MPix/s', r['value'])"; done; done; done
