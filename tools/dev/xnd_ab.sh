# GPU box: k_expiry_cat with one / four digits per convolution pass (-DDMZ_XND) against the shipped two.  usage: tools/dev/xnd_ab.sh
cd "$(dirname "$0")/../.."
for X in 1 4; do L=$(bash tools/dev/variant_lib.sh xnd$X expiry.hip -DDMZ_XND=$X) || echo "build failed $X"; done
for rep in 1 2 3; do for V in "" $PWD/gpurun_out/ab/lib_xnd1.so $PWD/gpurun_out/ab/lib_xnd4.so; do
  [ -n "$V" ] && [ ! -f "$V" ] && continue
  echo -n "${V:-shipped (2)}: "; DMZ_HIP_LIB=$V python tools/stage_times.py 65536 2 2>/dev/null | grep -o "expiry_cat [0-9.]*"
done; done
