for st in 2 1; do ODT_CONV_STAGES=$st python tools/profile_layers.py --batch 8 --steps 3 > gpurun_out/layers_st$st.txt 2>&1; done
for st in 2 1; do ODT_CONV_STAGES=$st python tools/profile_layers.py --batch 1 --steps 5 > gpurun_out/layers_b1_st$st.txt 2>&1; done
