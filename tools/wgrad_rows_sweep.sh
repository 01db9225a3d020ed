# Diagnostic: k_wgrad with other row ranges per work item (MMN_WGRAD_ROWS) / without the half-split of the 64 x 64 tiles
# (MMN_WGRAD_SPLIT=0); both are read when the plan is built
for w in c3 mimic; do for rows in 512 640 448 384; do MMN_WGRAD_ROWS=$rows python tools/time_kernels.py $w 2>/dev/null | tail -1 | sed "s/^/$w rows=$rows /"; done; done
MMN_WGRAD_SPLIT=0 python tools/time_kernels.py c3 2>/dev/null | tail -1 | sed "s/^/c3 nosplit /"
