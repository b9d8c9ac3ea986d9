# ---- sanitizer build of the HOST side (sanitizers on the CPU build only; GPU ASan is not available here).
# --offload-host-only compiles no device code at all, so this is the argument-validation / bookkeeping layer of every
# entry point under AddressSanitizer + UBSan; tests/abi/abi_driver.c drives it with invalid and extreme arguments
# (no call reaches a kernel launch, no GPU needed).  `make check-asan` builds and runs it.
ASAN_FLAGS = --offload-arch=$(ARCH) --offload-host-only -fsanitize=address,undefined -fno-sanitize-recover=undefined \
             -fno-omit-frame-pointer -O1 -g -std=c++17 -fPIC -Wall -Wno-unused-function
ASAN_OBJS = $(SRCS:%.hip=build-asan/%.o)
ASAN_LIB  = build-asan/librnamsm_hip_asan.so
ASAN_DRV  = build-asan/abi_driver

build-asan/%.o: %.hip common.h mma_core.h half16.h tile16.h row_split.h ../../include/rnamsm.h
	@mkdir -p build-asan
	$(HIPCC) $(ASAN_FLAGS) -c $< -o $@

# a host-only object still refers to the device code object it would embed (__hip_fatbin_<hash>): give every such symbol
# an empty stand-in, so the library links and loads; it can validate arguments but never launch
build-asan/fatbin_stubs.c: $(ASAN_OBJS)
	nm -u $(ASAN_OBJS) | grep -o '__hip_fatbin_[0-9a-f]*' | sort -u | \
	    sed 's/.*/const char &[4096] __attribute__((aligned(4096))) = {0};/' > $@

$(ASAN_LIB): $(ASAN_OBJS) build-asan/fatbin_stubs.c
	$(HIPCC) --offload-arch=$(ARCH) --offload-host-only -fsanitize=address,undefined -shared -fPIC -o $@ $(ASAN_OBJS) \
	    -x c build-asan/fatbin_stubs.c -x none

$(ASAN_DRV): ../../tests/abi/abi_driver.c $(ASAN_LIB) ../../include/rnamsm.h
	$(HIPCC) -x c -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -O1 -g \
	    ../../tests/abi/abi_driver.c -x none -o $@ -Lbuild-asan -lrnamsm_hip_asan -Wl,-rpath,'$$ORIGIN'

check-asan: $(ASAN_DRV)
	ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 ./$(ASAN_DRV)

