# Builds the HIP hot path (libsaev_amd.so) for gfx950.  hipcc cross-compiles without a GPU.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := saev_amd/csrc
SRCS  := $(CSRC)/ctx.hip $(CSRC)/gemm_encode.hip $(CSRC)/gemm_encode_f16x3.hip $(CSRC)/split.hip $(CSRC)/select.hip $(CSRC)/sparse.hip $(CSRC)/tail.hip $(CSRC)/auxk.hip
OBJS  := $(patsubst $(CSRC)/%.hip,build/%.o,$(SRCS))
FLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wno-pass-failed -Wno-unused-value -Iinclude

all: saev_amd/libsaev_amd.so

build/%.o: $(CSRC)/%.hip $(CSRC)/kernels.h $(CSRC)/common.h include/saev_amd.h
	@mkdir -p build
	$(HIPCC) $(FLAGS) -c $< -o $@

saev_amd/libsaev_amd.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@

# the inline-asm staging of gemm_encode_f16x3.hip owns m0: prove on the assembly that the compiler never relies on it
check-m0:
	HIPCC='$(HIPCC)' HIPFLAGS='$(FLAGS)' python3 tools/check_m0.py

clean:
	rm -rf build saev_amd/libsaev_amd.so

.PHONY: all clean check-m0
