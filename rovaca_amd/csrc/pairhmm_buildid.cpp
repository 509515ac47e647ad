// pairhmm_buildid.cpp -- the build id of the device code a library was linked from (see the Makefile: a hash of the kernel sources and
// of the compile flags, or whatever a variant build passes).  Compiled twice: once per library.
#if defined(PHMM_KERNEL_BUILD_ID)
extern "C" const char *phmm_kernel_build_id(void) { return PHMM_KERNEL_BUILD_ID; }
#endif
#if defined(SWHIP_KERNEL_BUILD_ID)
extern "C" const char *swhip_kernel_build_id(void) { return SWHIP_KERNEL_BUILD_ID; }
#endif
