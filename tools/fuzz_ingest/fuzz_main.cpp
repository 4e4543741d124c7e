#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
extern "C" {
int slsmi_audio_info(const char* path, int* sample_rate, int* channels, int64_t* frames);
int slsmi_audio_read(const char* path, float* out, int64_t capacity, int64_t* frames_out, int* sample_rate);
}
int main(int argc, char** argv) {
    int ok = 0, bad = 0;
    std::vector<float> buf(1 << 20);
    for (int i = 1; i < argc; ++i) {
        int sr = 0, ch = 0; int64_t fr = 0, got = 0;
        int rc1 = slsmi_audio_info(argv[i], &sr, &ch, &fr);
        int rc2 = slsmi_audio_read(argv[i], buf.data(), (int64_t)buf.size(), &got, &sr);
        if (rc1 == 0 && rc2 == 0) ++ok; else ++bad;
    }
    printf("ok %d rejected %d\n", ok, bad);
    return 0;
}
