# TEST INFRASTRUCTURE -- prints every `__global__` function DEFINITION of a CUDA source file (with the
# `template<...>` line in front of it, if any), from the `__global__` line up to the closing brace in
# column 0.  oracle/Makefile runs it over the reference's sort.cu / linear.cu WHERE THEY LIE and writes
# the output to the git-ignored oracle/_ref/*_kernels.inc, which the host shims (ref_sort_shim.cc,
# ref_linear_shim.cc) then #include -- so the kernel BODIES that get compiled are the reference's own
# text, none of it is stored in this repository, and the `<<<...>>>` host launchers (which no host
# compiler can parse) stay behind.
{
    if (infn) { print; if ($0 ~ /^}/) { infn = 0; print "" } next }
    if ($0 ~ /^template</) { held = $0; next }
    if ($0 ~ /^__global__/) { if (held != "") print held; held = ""; print; infn = 1; next }
    held = ""
}
