for abl in 0 8; do VQCPC_G3_ABL=$abl python tools/bench_g3_stagger.py child 2>&1 | grep stagger | sed "s/^/ABL=$abl /"; done
