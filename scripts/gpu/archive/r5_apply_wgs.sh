for v in stock bn_a2 bn_a8 bn_a16; do
  if [ $v = stock ]; then unset LIDAL_AMD_LIB; else export LIDAL_AMD_LIB=$PWD/scripts/_abl/lib_$v.so; fi
  echo "== $v"; python3 scripts/exp/bn_vs_torch.py 2>&1 | grep " x " | cut -c1-140
done
