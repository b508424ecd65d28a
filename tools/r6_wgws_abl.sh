for a in 0 1 2 3 8 16 32 35 0; do DVAE_WGWS_ABLATE=$a timeout 120 python tools/wgws_one.py 2>&1 | grep ABLATE; done
