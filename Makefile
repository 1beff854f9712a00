# Builds the C-ABI library without Python (what a Julia / C consumer needs):  make -j        -> rome.jl_amd/librome_mi355.so
#                                                                             make abi-smoke  -> tests/c/abi_smoke (needs a GPU to run)
#                                                                             make oracle     -> oracle/librome_oracle.so (test infrastructure)
# `python __graft_entry__.py` does the same through rome.jl_amd/_build.py.
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := rome.jl_amd/csrc
OBJDIR  := rome.jl_amd/build
SRC     := $(wildcard $(CSRC)/*.hip)
OBJ     := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(SRC))
HDR     := $(wildcard $(CSRC)/*.h $(CSRC)/*.hpp) include/rome_mi355.h
LIB     := rome.jl_amd/librome_mi355.so

$(LIB): $(OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ)

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wno-pass-failed -c $< -o $@

abi-smoke: $(LIB) tests/c/abi_smoke.c
	$(CC) -std=c11 -O2 -Iinclude -o tests/c/abi_smoke tests/c/abi_smoke.c -Lrome.jl_amd -lrome_mi355 -Wl,-rpath,'$$ORIGIN/../../rome.jl_amd' -lm

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(LIB) $(OBJDIR) tests/c/abi_smoke
	$(MAKE) -C oracle clean

.PHONY: abi-smoke oracle clean
