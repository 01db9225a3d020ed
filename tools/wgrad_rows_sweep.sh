# Diagnostic: k_wgrad with other row ranges per work item (MMN_WGRAD_ROWS, read when the plan is built)
for w in c3 mimic; do for rows in 512 448 384 352 320; do MMN_WGRAD_ROWS=$rows python tools/time_kernels.py $w 2>/dev/null | tail -1 | sed "s/^/$w rows=$rows /"; done; done
