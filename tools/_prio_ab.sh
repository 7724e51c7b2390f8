# A/B runs of fine-level Sinkhorn variants from diagnostic libraries (edit the suffix list): cost + OT per 20 224 problems
for i in 1 2 3; do python tools/fine_fused_time.py 20224 | sed "s/^/prod /"; for v in _wsum _xbar; do PATS_AMD_DIAG_LIB=$v python tools/fine_fused_time.py 20224 | sed "s/^/$v /"; done; done
