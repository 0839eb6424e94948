// hess_shared.hip -- the pinned result buffers of a context in node-shared memory (hess_share_results; see hess_ctx.h).
#include "hess_ctx.h"

namespace hess {

// A pinned result buffer of a context whose results are shared with other processes of the node (hess_share_results):
// a POSIX shared memory object, mapped and registered with the runtime, so that the copier's DMA copy (or the
// descriptor kernel's own stores) lands in memory the consumer process has mapped as well -- every GPU of a node
// delivers over its own host link and nothing is funnelled through one rank's.  `which` is 'k' or 'd'.
// Where /dev/shm has no room (containers often give it 64 MB) the buffer becomes a file under HESS_SHARE_DIR / TMPDIR /
// /tmp instead, mapped MAP_SHARED and registered the same way: page-cache pages, pinned by the registration -- the
// consumer maps the same pages.  Slower to set up, the same to use.  HESS_SHARE_FORCE_FILE=1 skips /dev/shm (tests).
int ensure_shared(hess_ctx* c, DevBuf& b, size_t bytes, char which) {
  if (bytes <= b.bytes) return 0;
  const long page = sysconf(_SC_PAGESIZE);
  size_t want = bytes + bytes / 4;  // grown by need: a quarter of slack so that batches of similar size do not reallocate
  want = (want + (size_t)page - 1) / (size_t)page * (size_t)page;
  uint32_t& gen = which == 'k' ? c->share_dir->gen_keys : c->share_dir->gen_desc;
  char name[256], path[1024];
  snprintf(name, sizeof(name), "/%s.%c%u", c->share.c_str(), which, gen + 1);
  // (posix_fallocate, not ftruncate: a full file system must fail here, not as a bus error at the first store)
  auto size_fd = [&](int fd) {
    int fe = posix_fallocate(fd, 0, (off_t)want);
    if (fe == EOPNOTSUPP || fe == EINVAL) fe = ftruncate(fd, (off_t)want) == 0 ? 0 : errno;
    return fe;
  };
  int fd = -1, why = 0;
  bool is_file = false;
  if (!dev_env("HESS_SHARE_FORCE_FILE")) {
    (void)shm_unlink(name);  // a stale object of a dead job with the same name
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) why = errno;
    else if ((why = size_fd(fd)) != 0) { close(fd); shm_unlink(name); fd = -1; }
    if (fd >= 0) snprintf(path, sizeof(path), "/dev/shm%s", name);
  } else {
    why = ENOSPC;
  }
  if (fd < 0) {  // the fallback: a file
    const char* dir = getenv("HESS_SHARE_DIR");
    if (!dir || !dir[0]) dir = getenv("TMPDIR");
    if (!dir || !dir[0]) dir = "/tmp";
    // the directory as an absolute path (a reader process may have another working directory), and a word of warning
    // when it is not memory-backed: the DMA copies of every batch then dirty page-cache pages the kernel writes to disk
    char absdir[PATH_MAX];
    if (realpath(dir, absdir)) dir = absdir;
    struct statfs sfs;
    if (statfs(dir, &sfs) == 0 && sfs.f_type != 0x01021994 /* TMPFS_MAGIC */ && sfs.f_type != 0x858458f6 /* RAMFS_MAGIC */ &&
        (c->p.verbose & 1))
      fprintf(stderr, "hessgpu: shared result buffer %s goes to %s, which is not a tmpfs: expect disk write-back per batch\n", name + 1, dir);
    snprintf(path, sizeof(path), "%s%s", dir, name);
    (void)unlink(path);
    fd = open(path, O_CREAT | O_EXCL | O_RDWR, 0600);
    int fe = fd < 0 ? errno : size_fd(fd);
    if (fe != 0) {
      if (fd >= 0) { close(fd); unlink(path); }
      set_err(c, "cannot place the shared result buffer %s (%zu bytes): /dev/shm: %s; %s: %s", name + 1, want, strerror(why), path, strerror(fe));
      return HESS_ERR_NOMEM;
    }
    is_file = true;
  }
  auto drop = [&]() { if (is_file) unlink(path); else shm_unlink(name); };
  void* np = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, 0);
  close(fd);
  if (np == MAP_FAILED) { set_err(c, "mmap(%s) failed: %s", path, strerror(errno)); drop(); return HESS_ERR_NOMEM; }
  void* dp = nullptr;
  hipError_t e = hipHostRegister(np, want, hipHostRegisterPortable | hipHostRegisterMapped);
  if (e == hipSuccess) e = hipHostGetDevicePointer(&dp, np, 0);
  if (e != hipSuccess || dp != np) {  // (the kernels and the copier address the buffer by its host pointer)
    if (e == hipSuccess) (void)hipHostUnregister(np);
    else (void)hipGetLastError();
    set_err(c, "cannot register the shared result buffer %s with the runtime: %s", path,
            e != hipSuccess ? hipGetErrorString(e) : "device alias differs from the host address");
    munmap(np, want); drop();
    return e == hipErrorOutOfMemory ? HESS_ERR_NOMEM : HESS_ERR_DEVICE;
  }
  release(b, true);
  try {
    b.shm = is_file ? path : name;
  } catch (...) {  // (nothing thrown crosses the C ABI)
    (void)hipHostUnregister(np);
    munmap(np, want); drop();
    set_err(c, "out of memory");
    return HESS_ERR_NOMEM;
  }
  b.p = np; b.bytes = want;
  // size and path first, the generation number last: a reader that sees the new generation sees its buffer
  (which == 'k' ? c->share_dir->keys_bytes : c->share_dir->desc_bytes) = want;
  snprintf(which == 'k' ? c->share_dir->keys_path : c->share_dir->desc_path, sizeof(c->share_dir->keys_path), "%s", path);
  __sync_synchronize();
  gen++;
  __sync_synchronize();
  return 0;
}


}  // namespace hess
