# A/B of the launch-shape knobs at small batch sizes (latency-path fused grids, auxiliary stream)
export SWEEP=${SWEEP:-128,96,64,48}
echo "== default"; python tools/prof/batch_size_sweep.py
echo "== SWF_LAT_FUSE_MAX=128"; SWF_LAT_FUSE_MAX=128 python tools/prof/batch_size_sweep.py
echo "== SWF_LAT_FUSE_MAX=128 SWF_AUX_STREAM_ALWAYS"; SWF_LAT_FUSE_MAX=128 SWF_AUX_STREAM_ALWAYS=1 python tools/prof/batch_size_sweep.py
echo "== SWF_AUX_STREAM_ALWAYS"; SWF_AUX_STREAM_ALWAYS=1 python tools/prof/batch_size_sweep.py
echo "== SWF_NO_AUX_STREAM"; SWF_NO_AUX_STREAM=1 python tools/prof/batch_size_sweep.py
