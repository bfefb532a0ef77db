# Host-side AddressSanitizer build: the HOST halves of every translation unit --
# argument checks, plan / launch arithmetic, workspace bookkeeping -- instrumented; device code is
# compiled as usual (GPU ASan needs xnack+, which this pool does not run).  `make asan`, then
# tests/test_abi.py::test_host_asan_build loads it under LD_PRELOAD of the ASan runtime.
ASAN_DIR = build_asan
ASAN_OBJS = $(addprefix $(ASAN_DIR)/,$(OBJS))
ASAN_FLAGS = -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer -g

asan: $(ASAN_DIR)/libsgp_amd_asan.so

$(ASAN_DIR)/libsgp_amd_asan.so: $(ASAN_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -fsanitize=address -o $@ $(ASAN_OBJS)

$(ASAN_DIR)/%.o: %.hip common.h reservoir_impl.h ../../include/sgp_amd.h
	@mkdir -p $(ASAN_DIR)
	$(HIPCC) $(CXXFLAGS) $(ASAN_FLAGS) -c $< -o $@
