for cfg in "100000 20 512 16" "100000 20 512 24" "50000 20 1024 10" "200000 10 128 10" "100000 10 512 4" "100000 10 256 20"; do
  set -- $cfg
  for mode in cols tiles; do
    GPARML_B_PHASE2=$mode python bench.py --steps 2 --warmup 1 --N $1 --D $2 --M $3 --Q $4 --regime B --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', '$mode', round(d['ms_per_step'],2), d['config']['device_ms']['p2_kernel_ms'])"
  done
done
