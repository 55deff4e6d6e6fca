// build: hipcc --offload-arch=gfx950 -O2 -o ipc_probe ipc_probe.cpp (run on the GPU box)
// profiles/probes/ipc_probe.cpp -- can a second process write into device memory of the first (hipIpc*)?
//   ipc_probe            parent: allocates, spawns itself as child with the handle, checks what the child wrote
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>
#include <sys/wait.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)
int main(int argc, char **argv) {
    if (argc == 2) {        // child: hex handle
        hipIpcMemHandle_t h;
        const char *hex = argv[1];
        for (size_t i = 0; i < sizeof(h); i++) { unsigned v; sscanf(hex + 2 * i, "%2x", &v); ((unsigned char *) &h)[i] = (unsigned char) v; }
        void *p = nullptr;
        CK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        CK(hipMemset(p, 0x5A, 1 << 20));
        CK(hipDeviceSynchronize());
        CK(hipIpcCloseMemHandle(p));
        return 0;
    }
    void *p = nullptr;
    CK(hipMalloc(&p, 1 << 20));
    CK(hipMemset(p, 0, 1 << 20));
    CK(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    CK(hipIpcGetMemHandle(&h, p));
    std::string hex;
    for (size_t i = 0; i < sizeof(h); i++) { char b[3]; snprintf(b, 3, "%02x", ((unsigned char *) &h)[i]); hex += b; }
    pid_t pid = fork();     // (exec at once: nothing of the parent's runtime is used in the child)
    if (pid == 0) { execl(argv[0], argv[0], hex.c_str(), (char *) nullptr); _exit(127); }
    int st = 0; waitpid(pid, &st, 0);
    unsigned char out[16];
    CK(hipMemcpy(out, p, 16, hipMemcpyDeviceToHost));
    printf("child exit %d, first byte 0x%02x -> %s\n", WEXITSTATUS(st), out[0], out[0] == 0x5A ? "IPC works" : "IPC does not work");
    return out[0] == 0x5A ? 0 : 1;
}
