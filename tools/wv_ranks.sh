# Dev tool: bisect which launches are not safe beside another engine of the same process (tools/debug_two_streams.py)
for cfg in "X=1" "BSMI_WINO=0" "BSMI_WINO=0 BSMI_H16=0 BSMI_FUSED_FIRST=0 BSMI_USE_BOX=0" "BSMI_WINO=0 BSMI_H16=0 BSMI_FUSED_FIRST=0 BSMI_USE_BOX=0 BSMI_SK_GRID=0" "BSMI_WINO=0 BSMI_H16=0 BSMI_FUSED_FIRST=0 BSMI_USE_BOX=0 BSMI_SK_GRID=0 BSMI_WAVES8=0" "BSMI_WINO=0 BSMI_H16=0 BSMI_FUSED_FIRST=0 BSMI_USE_BOX=0 BSMI_SK_GRID=0 BSMI_X3_FUSED=0" "PREC=bf16 BSMI_H16=0 BSMI_FUSED_FIRST=0 BSMI_USE_BOX=0 BSMI_SK_GRID=0 BSMI_USE_RH=0" "PREC=bf16 BSMI_H16=0 BSMI_FUSED_FIRST=0 BSMI_USE_BOX=0 BSMI_SK_GRID=0 BSMI_USE_RH=0 BSMI_WAVES8=0"; do
  echo "== $cfg"; env $cfg timeout -k 10 120 python tools/debug_two_streams.py 2>&1 | grep "differ\|rror" | tail -2
done
