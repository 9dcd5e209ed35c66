for lib in "" scratch/variants/lib_u2_b24.so scratch/variants/lib_u8_b24.so scratch/variants/lib_u16_b24.so scratch/variants/lib_u4_b16.so scratch/variants/lib_u4_b12.so; do
  for w in C2 C3; do V1C_LIB=${lib:+$PWD/$lib} python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${lib:-default(u4,b24)}', '$w', 'Gpx/s', round(d['value']/1e3,1), 'kernel_ms', d['roofline']['kernel_ms'])"; done
done
