for v in $VARIANTS; do
  cp ab/$v.so object_detection_tracking_amd/libodt_hip.so
  echo "#### $v"; python tools/conv_trace.py $SHAPES 2>&1 | grep -v "XCD0 first\|prologue split"
done
