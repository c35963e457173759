# Builds the product library (HIP, gfx950) and the test-side oracle.
HIPCC ?= hipcc
ARCH ?= gfx950
SRCS = zlib_amd/csrc/zamd_engine.hip zlib_amd/csrc/zamd_corpus.cpp zlib_amd/csrc/zamd_multi.cpp
HDRS = zlib_amd/csrc/zamd_core.h zlib_amd/csrc/zamd_inflate_impl.h zlib_amd/csrc/zamd_fast_impl.h zlib_amd/csrc/zamd_follow_impl.h include/zamd.h

all: zlib_amd/libzamd.so zlib_amd/libzamd_z.so examples/zamd_gzip oracle

zlib_amd/libzamd.so: $(SRCS) $(HDRS)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -pthread -Wno-unused-value -Wno-unused-result -o $@ $(SRCS) -ldl

# The reference's symbol versions (programs linked against a versioned libz find their symbols when this library is
# preloaded or linked in its place).  The SONAME stays libzamd_z.so on purpose: under the name libz.so.1 the loader would
# also hand this library to ROCm's own dependencies (libelf needs libz.so.1), whose inflate() calls during HSA start-up
# would then re-enter a library that is waiting for that very start-up.
zlib_amd/libzamd_z.so: zlib_amd/csrc/zamd_zlib.cpp zlib_amd/csrc/zamd_gz.cpp zlib_amd/csrc/zamd_z.map include/zamd_zlib.h include/zamd.h zlib_amd/libzamd.so
	g++ -O2 -std=c++17 -fPIC -shared -pthread -o $@ zlib_amd/csrc/zamd_zlib.cpp zlib_amd/csrc/zamd_gz.cpp -Lzlib_amd -lzamd -Wl,-Bsymbolic -Wl,-rpath,'$$ORIGIN' \
	    -Wl,-soname,libzamd_z.so -Wl,--version-script=zlib_amd/csrc/zamd_z.map

examples/zamd_gzip: examples/zamd_gzip.c include/zamd_zlib.h zlib_amd/libzamd_z.so
	gcc -O2 -Iinclude -o $@ examples/zamd_gzip.c -Lzlib_amd -lzamd_z -lzamd -Wl,-rpath,'$$ORIGIN/../zlib_amd'

oracle:
	$(MAKE) -C oracle

tests/host_emul: tests/host_emul.cpp zlib_amd/csrc/zamd_core.h oracle/zoracle.c
	gcc -O2 -c -o /tmp/zoracle_emul.o oracle/zoracle.c
	g++ -O2 -std=c++17 -I zlib_amd/csrc -I oracle -o $@ tests/host_emul.cpp /tmp/zoracle_emul.o

clean:
	rm -f zlib_amd/libzamd.so zlib_amd/libzamd_z.so tests/host_emul
.PHONY: all oracle clean

# the parse-following matcher with its counters compiled in (tests/gpu_spec_stats.py)
zlib_amd/libzamd_stats.so: $(SRCS) $(HDRS)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -shared -pthread -Wno-unused-value -Wno-unused-result -DZAMD_SPEC_STATS -o $@ $(SRCS) -ldl
