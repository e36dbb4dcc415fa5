# Builds the C-ABI shared library of the hot path for gfx950 (MI355X), in-tree.
#   make -j8        -> g_adaptivity_amd/libgadapt_hip.so   (one object per kernel family: csrc/gadapt_internal.h)
#   make resources  -> per-kernel VGPR/SGPR/LDS/occupancy report
#   make DEV_C=64   -> development build: tiled kernels for one hidden size only (never shipped)
HIPCC      ?= /opt/rocm/bin/hipcc
ARCH       ?= gfx950
CSRC       := g_adaptivity_amd/csrc
OBJDIR     := build/obj
LIB        := g_adaptivity_amd/libgadapt_hip.so
HIPFLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Iinclude -I$(CSRC) -Wall -Wno-unused-variable -Wno-unused-but-set-variable -Wno-unused-function $(EXTRA)
ifdef DEV_C
HIPFLAGS   += -DGADAPT_DEV_C=$(DEV_C)
endif
UNITS      := gadapt_kernels gadapt_tu_fwd gadapt_tu_bwd_target gadapt_tu_bwd_source gadapt_tu_smallmesh gadapt_tu_sparse gadapt_tu_gat
OBJS       := $(UNITS:%=$(OBJDIR)/%.o) $(OBJDIR)/csr_build.o
SHARED     := $(CSRC)/gadapt_internal.h $(CSRC)/gadapt_common.inc include/gadapt_hip.h

all: $(LIB)

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

$(OBJDIR)/gadapt_kernels.o: $(CSRC)/gadapt_small.inc
$(OBJDIR)/gadapt_tu_fwd.o: $(CSRC)/gadapt_fwd.inc $(CSRC)/gadapt_wide.inc
$(OBJDIR)/gadapt_tu_bwd_target.o: $(CSRC)/gadapt_bwd_target.inc
$(OBJDIR)/gadapt_tu_bwd_source.o: $(CSRC)/gadapt_bwd_source.inc
$(OBJDIR)/gadapt_tu_smallmesh.o: $(CSRC)/gadapt_smallmesh.inc
$(OBJDIR)/gadapt_tu_sparse.o: $(CSRC)/gadapt_sparse.inc
$(OBJDIR)/gadapt_tu_gat.o: $(CSRC)/gadapt_gat.inc

# FLAGS_<unit>="-D..." adds flags to ONE translation unit (A/B builds of a kernel family: tools/ab_units.sh)
$(OBJDIR)/%.o: $(CSRC)/%.hip $(SHARED)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(FLAGS_$*) -c -o $@ $<

$(OBJDIR)/csr_build.o: $(CSRC)/csr_build.cpp include/gadapt_hip.h
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<

resources:
	python3 tools/resources.py

clean:
	rm -rf $(OBJDIR) $(LIB)

.PHONY: all resources clean
