"""Build an A/B variant of the library: python tools/build_variant.py <gemm1x1-variant.hip> <out.so>
(every other object comes from the in-tree build; select at run time with DEEPHAR_HIP_LIB=<out.so>)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephar_amd.csrc import build as B
B.build(verbose=False)
src, out = sys.argv[1], sys.argv[2]
name = os.path.basename(src)
target = [s for s in B.SOURCES if name.startswith(s.replace('.hip', ''))][0]
obj = '/tmp/variant_%s.o' % os.path.basename(out)
subprocess.run([B.HIPCC] + B.FLAGS + ['-c', src, '-o', obj], check=True, capture_output=True)
objs = [obj if s == target else os.path.join(B.OBJDIR, s.replace('.hip', '.o')) for s in B.SOURCES]
subprocess.run([B.HIPCC, '--offload-arch=' + B.ARCH, '-shared', '-fPIC', '-o', out] + objs, check=True)
print('built', out)
