for cfg in cfg2_64ch_256taps cfg2_64ch_512taps; do
for k in "" "MFM_FORCE_DOT2=1"; do
echo -n "$cfg $k: "; env $k timeout 250 python bench.py --config $cfg --steps 60 --warmup 60 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], d['roofline']['kernel_ms'], d['value'], d['compute_roofline']['frac'])"
done; done
# BASELINE configs[4] per-GPU share: 10 MS/s, D = 400, 512 taps, 256 channels (single-iteration tiles, streamed taps)
for k in "" "MFM_FORCE_DOT2=1"; do
echo -n "cfg5_airspy 256ch $k: "; env $k timeout 250 python bench.py --config cfg5_airspy --channels-per-gpu 256 --steps 30 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel'], d['roofline']['kernel_ms'], d['value'], d['compute_roofline']['frac'])"
done
