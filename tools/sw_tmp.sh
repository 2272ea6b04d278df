for TB in 64 128 256; do for W in 3 4 6 8 12; do
echo -n "c2 TB=$TB WG_PER_CU=$W : "; JPEGGPU_TARGET_BLOCKS=$TB JPEGGPU_WG_PER_CU=$W timeout -k 10 120 python3 bench.py --steps 100 --no-cpu-baseline --workload c2 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['roofline']['achieved'], 'GB/s', r['roofline']['avg_launch_us'])"
done; done
