# The instruction schedulers pairhmm_kernels_tab.hip is compiled with, one build each (-DPHMM_TAB_SCHED=1, 2; build 0 takes hipcc's
# default): which class goes to which build is tab_plan in pairhmm_systolic_tab.h.  Part of the kernel build id (Makefile, KERNEL_SRC).
TAB_SCHED_1 := -mllvm -amdgpu-sched-strategy=max-ilp
TAB_SCHED_2 := -mllvm -amdgpu-sched-strategy=iterative-ilp
