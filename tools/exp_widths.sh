mkdir -p gpurun_out/r03h
run() { # tag, env..., lib, classes
  tag=$1; shift
  env "$@" timeout 600 python tools/perf_classes.py 3 600000 $CLS > gpurun_out/r03h/$tag.log 2>&1
  echo "== $tag"; cut -c1-110 gpurun_out/r03h/$tag.log | grep class
}
CLS="ABCD+" run base X=1
CLS="A+" run ev32 BTGPU_LIB=bayestyper_amd/libbtgpu_ev32.so
CLS="A+" run refill BTGPU_LIB=bayestyper_amd/libbtgpu_refill.so
CLS="A+" run both BTGPU_LIB=bayestyper_amd/libbtgpu_both.so
CLS="BCD+" run tail8 BT_GIBBS_TAIL_WIDTH=8
CLS="BCD+" run tail16 BT_GIBBS_TAIL_WIDTH=16
CLS="B+" run mid32 BT_GIBBS_MID_WIDTH=32
CLS="B+" run mid8 BT_GIBBS_MID_WIDTH=8
