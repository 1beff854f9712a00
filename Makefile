# Builds the C-ABI library without Python (what a Julia / C consumer needs):  make            -> rome.jl_amd/librome_mi355.so
#                                                                             make abi-smoke  -> tests/c/abi_smoke (needs a GPU to run)
#                                                                             make oracle     -> oracle/librome_oracle.so (test infrastructure)
# `python __graft_entry__.py` does the same through rome.jl_amd/_build.py.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := rome.jl_amd/csrc
SRC     := $(CSRC)/rome_kernels.hip $(CSRC)/rome_parametric.hip $(CSRC)/rome_product.hip $(CSRC)/rome_kde.hip $(CSRC)/rome_capi.hip
DEPS    := $(SRC) $(CSRC)/rome_kernels.h $(CSRC)/rome_device_math.hpp include/rome_mi355.h
LIB     := rome.jl_amd/librome_mi355.so

$(LIB): $(DEPS)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -Wno-pass-failed -o $@ $(SRC)

abi-smoke: $(LIB) tests/c/abi_smoke.c
	$(CC) -std=c11 -O2 -Iinclude -o tests/c/abi_smoke tests/c/abi_smoke.c -Lrome.jl_amd -lrome_mi355 -Wl,-rpath,'$$ORIGIN/../../rome.jl_amd' -lm

oracle:
	$(MAKE) -C oracle

clean:
	rm -f $(LIB) tests/c/abi_smoke
	$(MAKE) -C oracle clean

.PHONY: abi-smoke oracle clean
