# Builds the C-ABI shared library of the hot path for gfx950 (MI355X), in-tree.
#   make            -> g_adaptivity_amd/libgadapt_hip.so
#   make resources  -> per-kernel VGPR/SGPR/LDS/occupancy report
HIPCC      ?= /opt/rocm/bin/hipcc
ARCH       ?= gfx950
CSRC       := g_adaptivity_amd/csrc
LIB        := g_adaptivity_amd/libgadapt_hip.so
HIPFLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -Wall -Wno-unused-variable -Wno-unused-but-set-variable

all: $(LIB)

$(LIB): $(CSRC)/gadapt_kernels.hip $(CSRC)/gadapt_wide.inc $(CSRC)/gadapt_sparse.inc $(CSRC)/csr_build.cpp include/gadapt_hip.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/gadapt_kernels.hip $(CSRC)/csr_build.cpp

resources: $(CSRC)/gadapt_kernels.hip include/gadapt_hip.h
	$(HIPCC) $(HIPFLAGS) -c -o /dev/null $(CSRC)/gadapt_kernels.hip -Rpass-analysis=kernel-resource-usage 2>&1 | \
	  grep -E "Function Name|VGPRs:|AGPRs|SGPRs:|Occupancy|LDS Size|ScratchSize" | paste - - - - - - - | sed 's/remark: [^:]*:[0-9]*:[0-9]*: //g'

clean:
	rm -f $(LIB)

.PHONY: all resources clean
