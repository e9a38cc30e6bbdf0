#!/bin/bash
# Config 5 on one card: both stand-in streams, one rank and two child ranks sharing the card (gloo for the scalar reductions).
# usage (GPU box): bash tools/r03_chunk_stream.sh > gpurun_out/r03_chunk_stream.txt
set -e
export QSAE_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
TMP=$(mktemp -d)
for mode in gauss heavy; do
  python tools/run_chunk_stream.py --make $TMP/$mode.pt --mode $mode --contexts 2100
  echo "== $mode, 1 rank (2 batches in flight), with list statistics of the first batch"
  python tools/run_chunk_stream.py --chunk $TMP/$mode.pt --mode $mode --stats
  echo "== $mode, 1 rank, blocking forward_compact (1 batch in flight)"
  python tools/run_chunk_stream.py --chunk $TMP/$mode.pt --mode $mode --in-flight 1
  echo "== $mode, 2 ranks on the one card"
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
      tools/run_chunk_stream.py --chunk $TMP/$mode.pt --mode $mode 2>/dev/null | grep '^{'
done
rm -rf $TMP
