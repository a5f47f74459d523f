"""Build libcufhe_amd_var_<name>.so with extra -D switches (experiments on kernel variants; never loaded by the package):
   python tools/build_variant.py pf3 -DCUFHE_AMD_Q_PF3        then   CUFHE_AMD_LIBRARY=cufhe_amd/libcufhe_amd_var_pf3.so python tools/lvl2_ab.py"""
import importlib.util
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "cufhe_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
name, flags = sys.argv[1], sys.argv[2:]
print(b._compile_and_link(os.path.join(ROOT, "cufhe_amd", f"libcufhe_amd_var_{name}.so"), flags, (), "-v" in flags))
