set -e
export QSAE_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
TMP=$(mktemp -d)
python tools/run_chunk_stream.py --make $TMP/heavy.pt --mode heavy --contexts 2100
echo "== heavy, 4 ranks on the one card (gloo for the scalar reductions)"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29519 tools/run_chunk_stream.py --chunk $TMP/heavy.pt --mode heavy 2>/dev/null | grep '^{'
rm -rf $TMP
