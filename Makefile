# Builds the C-ABI shared library of the hot path for gfx950 (MI355X), in-tree.
#   make            -> g_adaptivity_amd/libgadapt_hip.so
#   make resources  -> per-kernel VGPR/SGPR/LDS/occupancy report
HIPCC      ?= /opt/rocm/bin/hipcc
ARCH       ?= gfx950
CSRC       := g_adaptivity_amd/csrc
LIB        := g_adaptivity_amd/libgadapt_hip.so
HIPFLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -Wall -Wno-unused-variable -Wno-unused-but-set-variable

all: $(LIB)

$(LIB): $(CSRC)/gadapt_kernels.hip $(wildcard $(CSRC)/*.inc) $(CSRC)/csr_build.cpp include/gadapt_hip.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/gadapt_kernels.hip $(CSRC)/csr_build.cpp

resources:
	python3 tools/resources.py

clean:
	rm -f $(LIB)

.PHONY: all resources clean
