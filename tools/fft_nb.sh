for nb in 1 2; do for g in 2560 5120 10240; do echo "NB $nb"; MKT_FFT_NB=$nb GRIDS=$g bash tools/fft_bench.sh; done; done
