# usage: bash tools/tile_sweep.sh   (tuning aid: direct vs Winograd tiles on the main C2/C4 3x3 shapes)
for shape in "64 64 3 16 128 256" "128 64 3 16 128 256" "64 64 3 64 64 128"; do
echo -n "$shape direct: "; python tools/one_conv.py $shape 20 2>/dev/null
for t in 7 10; do echo -n "$shape winograd tile=$t: "; SDC_PRECISION=2 SDC_WG_TILE=$t python tools/one_conv.py $shape 20 2>/dev/null; done; done
