# usage (on the GPU box): bash tools/switch_matrix.sh  -> one line per fallback switch of round 3: 51 parity tests under it
for v in UMPR_MERGE_SMALL=0 UMPR_MERGE_DX=1 UMPR_B16_POOL_BWD_WIN=0 UMPR_WINO_BIAS_FUSE=0 UMPR_FC_K32=0 UMPR_EMB_GATHER=0 UMPR_WINO_C21_FWD=0 UMPR_GRU_V1=1 UMPR_TEXT_STREAM=0 UMPR_WGRAD_STREAM=0; do
  env $v UMPR_TEST_CHILD=1 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py -q -m gpu -p no:cacheprovider -k "golden or review_head or control or embed_gru or umpr_r_small or trained or eval_mse or maxpool_bf16 or vgg16_small or classifier_at" > gpurun_out/matrix_$v.log 2>&1
  echo "$v: $(tail -1 gpurun_out/matrix_$v.log)"
done
