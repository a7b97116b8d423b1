# usage: bash tools/tile_sweep.sh   (tuning aid: direct form vs every Winograd tile on the main C2 / C4 3x3 shapes)
# SDC_WG_TILE: 3 = 64x128 (4 waves), 6 = 128x128, 7 = 64x256, 9 = 128x256, 10 = 64x512 (8 waves)
for shape in "64 64 3 16 128 256" "128 64 3 16 128 256" "128 128 3 8 64 256" "256 256 3 4 32 256" "512 512 3 2 16 256" "64 64 3 64 64 128"; do
echo -n "$shape direct: "; python tools/one_conv.py $shape 20 2>/dev/null
for t in 3 6 7 9 10; do echo -n "$shape winograd tile=$t: "; SDC_PRECISION=2 SDC_WG_TILE=$t python tools/one_conv.py $shape 20 2>/dev/null; done; done
