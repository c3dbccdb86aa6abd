for f in 0 256; do for lb in 64 16 8 4; do for at in 0 300 600; do
  SDRPP_GPU_FCM16_MAX_TILES=$f SDRPP_GPU_TICK_LAND_BLOCKS=$lb SDRPP_GPU_TICK_L0_AT=$at python tools/land_sweep.py 2>/dev/null
done; done; done
