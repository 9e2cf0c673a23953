mkdir -p gpurun_out
for u in 2 3 4 6 8; do python3 tools/time_step.py C2 C4 C3 --marg 4 4 --tune marg_piece_units=$u; done 2>&1 | tee gpurun_out/sweep_piece.txt
