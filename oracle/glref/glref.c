/* glref -- the reference's OWN shaders, executed.  TEST INFRASTRUCTURE ONLY (oracle/): never linked or loaded by the product.
 *
 * hyperlogic/splatapult's hot path is GLSL 4.60 behind an OpenGL driver; SURVEY.md 8c judged it unrunnable here (no display, no
 * EGL / OSMesa).  The image does carry Mesa 23.2's software rasteriser (swrast_dri.so = llvmpipe), and its DRI "swrast" interface
 * gives an OpenGL 4.5 core context (4.6 / GLSL 4.60 with Mesa's version overrides) without any window system: this file is that
 * loader plus the handful of GL calls SplatRenderer makes, so that the reference's shader FILES -- read at run time from
 * /root/reference/shader where they lie, never copied -- compute keys and pixels that pin oracle/msplat_oracle.c:
 *   presort_compute.glsl                          <- SplatRenderer::Sort's "pre-sort"   (src/splatrenderer.cpp:171-193)
 *   splat_vert.glsl + splat_geom.glsl + splat_frag.glsl, GL_POINTS through the real rasteriser and blender
 *                                                 <- SplatRenderer::Render              (src/splatrenderer.cpp:315-343)
 * Mirrored host behaviour, each with its reference line: the HEADER / DEFINES macro expansion (src/core/program.cpp:33-49,114,
 * 123-129; FULL_SH / FRAMEBUFFER_SRGB defines src/splatrenderer.cpp:60-72), the vertex layout of BuildVertexArrayObject (:345-391:
 * attributes bound by NAME to the interleaved AoS), the uniforms of Sort / Render (:175-177, :327-335), the blend / clear state of
 * App's Clear() (src/app.cpp:144-164) and the colour-only float FBO of --fp32 (:1018-1027).  The sort itself is not run
 * (KHR_shader_subgroup is absent on llvmpipe, where the reference falls back to the vendored sorter, :86): the draw order is an
 * input, like the element buffer the reference fills at :296-311.
 * Built by oracle/Makefile into oracle/_ref/libglref.so; used by tests/ and tests/golden/make_glref_golden.py only. */
#include <GL/internal/dri_interface.h>
#include <GL/glcorearb.h>
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static char g_err[4096];
static int fail(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}
const char* glref_last_error(void) { return g_err; }

/* ---- DRI swrast loader: the callbacks a window system would supply; the drawable is a dummy, all rendering goes to FBOs ---- */
static void get_drawable_info(__DRIdrawable* d, int* x, int* y, int* w, int* h, void* p) { (void)d; (void)p; *x = *y = 0; *w = *h = 16; }
static void put_image(__DRIdrawable* d, int op, int x, int y, int w, int h, char* data, void* p) { (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void get_image(__DRIdrawable* d, int x, int y, int w, int h, char* data, void* p) { (void)d; (void)x; (void)y; (void)p; memset(data, 0, (size_t)w * h * 4); }
static void put_image2(__DRIdrawable* d, int op, int x, int y, int w, int h, int stride, char* data, void* p) { (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p; }
static void get_image2(__DRIdrawable* d, int x, int y, int w, int h, int stride, char* data, void* p) { (void)d; (void)x; (void)y; (void)w; (void)p; memset(data, 0, (size_t)stride * h); }
static const __DRIswrastLoaderExtension swrast_loader = {
    .base = {__DRI_SWRAST_LOADER, 3},
    .getDrawableInfo = get_drawable_info, .putImage = put_image, .getImage = get_image, .putImage2 = put_image2, .getImage2 = get_image2,
};
static const __DRIextension* loader_exts[] = {&swrast_loader.base, NULL};

#define GLF(type, name) static type p##name
GLF(PFNGLGETSTRINGPROC, glGetString); GLF(PFNGLGETERRORPROC, glGetError);
GLF(PFNGLCREATESHADERPROC, glCreateShader); GLF(PFNGLSHADERSOURCEPROC, glShaderSource); GLF(PFNGLCOMPILESHADERPROC, glCompileShader);
GLF(PFNGLGETSHADERIVPROC, glGetShaderiv); GLF(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog); GLF(PFNGLCREATEPROGRAMPROC, glCreateProgram);
GLF(PFNGLATTACHSHADERPROC, glAttachShader); GLF(PFNGLLINKPROGRAMPROC, glLinkProgram); GLF(PFNGLGETPROGRAMIVPROC, glGetProgramiv);
GLF(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog); GLF(PFNGLUSEPROGRAMPROC, glUseProgram); GLF(PFNGLGETUNIFORMLOCATIONPROC, glGetUniformLocation);
GLF(PFNGLUNIFORMMATRIX4FVPROC, glUniformMatrix4fv); GLF(PFNGLUNIFORM2FVPROC, glUniform2fv); GLF(PFNGLUNIFORM3FVPROC, glUniform3fv);
GLF(PFNGLUNIFORM4FVPROC, glUniform4fv); GLF(PFNGLUNIFORM1UIPROC, glUniform1ui); GLF(PFNGLGENBUFFERSPROC, glGenBuffers);
GLF(PFNGLBINDBUFFERPROC, glBindBuffer); GLF(PFNGLBUFFERDATAPROC, glBufferData); GLF(PFNGLBINDBUFFERBASEPROC, glBindBufferBase);
GLF(PFNGLGETBUFFERSUBDATAPROC, glGetBufferSubData); GLF(PFNGLDELETEBUFFERSPROC, glDeleteBuffers); GLF(PFNGLDISPATCHCOMPUTEPROC, glDispatchCompute);
GLF(PFNGLMEMORYBARRIERPROC, glMemoryBarrier); GLF(PFNGLFINISHPROC, glFinish); GLF(PFNGLGENVERTEXARRAYSPROC, glGenVertexArrays);
GLF(PFNGLBINDVERTEXARRAYPROC, glBindVertexArray); GLF(PFNGLDELETEVERTEXARRAYSPROC, glDeleteVertexArrays); GLF(PFNGLGETATTRIBLOCATIONPROC, glGetAttribLocation);
GLF(PFNGLENABLEVERTEXATTRIBARRAYPROC, glEnableVertexAttribArray); GLF(PFNGLVERTEXATTRIBPOINTERPROC, glVertexAttribPointer);
GLF(PFNGLGENTEXTURESPROC, glGenTextures); GLF(PFNGLBINDTEXTUREPROC, glBindTexture); GLF(PFNGLTEXIMAGE2DPROC, glTexImage2D);
GLF(PFNGLTEXPARAMETERIPROC, glTexParameteri); GLF(PFNGLDELETETEXTURESPROC, glDeleteTextures); GLF(PFNGLGENFRAMEBUFFERSPROC, glGenFramebuffers);
GLF(PFNGLBINDFRAMEBUFFERPROC, glBindFramebuffer); GLF(PFNGLFRAMEBUFFERTEXTURE2DPROC, glFramebufferTexture2D);
GLF(PFNGLCHECKFRAMEBUFFERSTATUSPROC, glCheckFramebufferStatus); GLF(PFNGLDELETEFRAMEBUFFERSPROC, glDeleteFramebuffers);
GLF(PFNGLVIEWPORTPROC, glViewport); GLF(PFNGLENABLEPROC, glEnable); GLF(PFNGLDISABLEPROC, glDisable); GLF(PFNGLBLENDEQUATIONPROC, glBlendEquation);
GLF(PFNGLBLENDFUNCPROC, glBlendFunc); GLF(PFNGLCLEARCOLORPROC, glClearColor); GLF(PFNGLCLEARPROC, glClear); GLF(PFNGLDRAWELEMENTSPROC, glDrawElements);
GLF(PFNGLREADPIXELSPROC, glReadPixels); GLF(PFNGLPIXELSTOREIPROC, glPixelStorei);
GLF(PFNGLGENRENDERBUFFERSPROC, glGenRenderbuffers); GLF(PFNGLBINDRENDERBUFFERPROC, glBindRenderbuffer);
GLF(PFNGLRENDERBUFFERSTORAGEPROC, glRenderbufferStorage); GLF(PFNGLFRAMEBUFFERRENDERBUFFERPROC, glFramebufferRenderbuffer);
GLF(PFNGLDELETERENDERBUFFERSPROC, glDeleteRenderbuffers);
GLF(PFNGLGENERATEMIPMAPPROC, glGenerateMipmap); GLF(PFNGLACTIVETEXTUREPROC, glActiveTexture); GLF(PFNGLUNIFORM1IPROC, glUniform1i);
GLF(PFNGLUNIFORM1FPROC, glUniform1f); GLF(PFNGLGETTEXIMAGEPROC, glGetTexImage);

static int g_ready = 0, g_full_sh = 0;
static GLuint g_presort = 0, g_splat = 0, g_point = 0;
static GLuint g_splat_cfg[2][2];          /* [full_sh][srgb]: the splat program per pair of defines, compiled on first use */
static char g_version[256], g_dir[1024];

static char* read_file(const char* dir, const char* name)
{
    char path[1024];
    snprintf(path, sizeof(path), "%s/%s", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) { fail("cannot open %s", path); return NULL; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* s = (char*)malloc((size_t)n + 1);
    if (fread(s, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(s); fail("short read of %s", path); return NULL; }
    s[n] = 0;
    fclose(f);
    return s;
}

/* Program::ExpandMacros (src/core/program.cpp:33-49): every occurrence of the token is replaced by the value */
static char* expand(char* src, const char* token, const char* value)
{
    size_t tl = strlen(token), vl = strlen(value), cap = strlen(src) + 1;
    for (char* p = src; (p = strstr(p, token)); p += tl) cap += vl;
    char* out = (char*)malloc(cap + 1);
    char *o = out, *p = src, *q;
    while ((q = strstr(p, token))) {
        memcpy(o, p, (size_t)(q - p)); o += q - p;
        memcpy(o, value, vl); o += vl;
        p = q + tl;
    }
    strcpy(o, p);
    free(src);
    return out;
}

static GLuint compile(GLenum type, const char* dir, const char* name, const char* defines)
{
    char* src = read_file(dir, name);
    if (!src) return 0;
    src = expand(src, "/*%%HEADER%%*/", "#version 460");            /* Program::Program(), src/core/program.cpp:114 */
    src = expand(src, "/*%%DEFINES%%*/", defines);                   /* splatrenderer.cpp:60-72 */
    GLuint sh = pglCreateShader(type);
    const char* srcs[1] = {src};
    pglShaderSource(sh, 1, srcs, NULL);
    pglCompileShader(sh);
    GLint ok = 0;
    pglGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
    free(src);
    if (!ok) {
        char log[3000];
        pglGetShaderInfoLog(sh, sizeof(log), NULL, log);
        fail("compile %s: %s", name, log);
        return 0;
    }
    return sh;
}

static GLuint link_program(GLuint a, GLuint b, GLuint c)
{
    GLuint p = pglCreateProgram();
    if (a) pglAttachShader(p, a);
    if (b) pglAttachShader(p, b);
    if (c) pglAttachShader(p, c);
    pglLinkProgram(p);
    GLint ok = 0;
    pglGetProgramiv(p, GL_LINK_STATUS, &ok);
    if (!ok) {
        char log[3000];
        pglGetProgramInfoLog(p, sizeof(log), NULL, log);
        fail("link: %s", log);
        return 0;
    }
    return p;
}

static int glref_configure(int full_sh, int srgb);

/* shader_dir: the reference's shader directory (/root/reference/shader).  full_sh / srgb: the FULL_SH / FRAMEBUFFER_SRGB defines of
 * SplatRenderer::Init (splatrenderer.cpp:60-72); calling it again with other defines switches the splat program. */
int glref_init(const char* shader_dir, int full_sh, int srgb)
{
    if (g_ready) return glref_configure(full_sh, srgb);
    snprintf(g_dir, sizeof(g_dir), "%s", shader_dir);
    /* the shaders say "#version 460"; llvmpipe of this Mesa advertises 4.5 and implements what they use */
    setenv("MESA_GL_VERSION_OVERRIDE", "4.6", 1);
    setenv("MESA_GLSL_VERSION_OVERRIDE", "460", 1);
    /* texture sampling (the point sprites only; the splat shaders sample nothing): float filtering and per-pixel level of detail
     * instead of llvmpipe's 8-bit fixed-point filter weights and per-quad LOD -- the most exact form this GL offers */
    setenv("GALLIVM_PERF", "no_aos_sampling,no_quad_lod", 0);
    void* drv = dlopen("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
    if (!drv) return fail("dlopen swrast_dri.so: %s", dlerror());
    const __DRIextension** (*get_exts)(void) = (const __DRIextension** (*)(void))dlsym(drv, "__driDriverGetExtensions_swrast");
    if (!get_exts) return fail("swrast_dri.so has no __driDriverGetExtensions_swrast");
    const __DRIextension** exts = get_exts();
    const __DRIcoreExtension* core = NULL;
    const __DRIswrastExtension* swrast = NULL;
    for (int i = 0; exts[i]; ++i) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension*)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) swrast = (const __DRIswrastExtension*)exts[i];
    }
    if (!core || !swrast || swrast->base.version < 4) return fail("DRI core / swrast (v4) extension missing");
    const __DRIconfig** configs = NULL;
    __DRIscreen* screen = swrast->createNewScreen2(0, loader_exts, exts, &configs, NULL);
    if (!screen || !configs || !configs[0]) return fail("createNewScreen2 failed");
    unsigned err = 0;
    const uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 6};
    __DRIcontext* ctx = swrast->createContextAttribs(screen, __DRI_API_OPENGL_CORE, configs[0], NULL, 2, attribs, &err, NULL);
    if (!ctx) return fail("createContextAttribs (OpenGL 4.6 core) failed: %u", err);
    __DRIdrawable* dr = swrast->createNewDrawable(screen, configs[0], NULL);
    if (!dr || !core->bindContext(ctx, dr, dr)) return fail("createNewDrawable / bindContext failed");
    void* glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!glapi) return fail("dlopen libglapi.so.0: %s", dlerror());
    void* (*gpa)(const char*) = (void* (*)(const char*))dlsym(glapi, "_glapi_get_proc_address");
    if (!gpa) return fail("no _glapi_get_proc_address");
#define LOAD(type, name) do { p##name = (type)gpa(#name); if (!p##name) return fail("GL entry point %s missing", #name); } while (0)
    LOAD(PFNGLGETSTRINGPROC, glGetString); LOAD(PFNGLGETERRORPROC, glGetError);
    LOAD(PFNGLCREATESHADERPROC, glCreateShader); LOAD(PFNGLSHADERSOURCEPROC, glShaderSource); LOAD(PFNGLCOMPILESHADERPROC, glCompileShader);
    LOAD(PFNGLGETSHADERIVPROC, glGetShaderiv); LOAD(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog); LOAD(PFNGLCREATEPROGRAMPROC, glCreateProgram);
    LOAD(PFNGLATTACHSHADERPROC, glAttachShader); LOAD(PFNGLLINKPROGRAMPROC, glLinkProgram); LOAD(PFNGLGETPROGRAMIVPROC, glGetProgramiv);
    LOAD(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog); LOAD(PFNGLUSEPROGRAMPROC, glUseProgram); LOAD(PFNGLGETUNIFORMLOCATIONPROC, glGetUniformLocation);
    LOAD(PFNGLUNIFORMMATRIX4FVPROC, glUniformMatrix4fv); LOAD(PFNGLUNIFORM2FVPROC, glUniform2fv); LOAD(PFNGLUNIFORM3FVPROC, glUniform3fv);
    LOAD(PFNGLUNIFORM4FVPROC, glUniform4fv); LOAD(PFNGLUNIFORM1UIPROC, glUniform1ui); LOAD(PFNGLGENBUFFERSPROC, glGenBuffers);
    LOAD(PFNGLBINDBUFFERPROC, glBindBuffer); LOAD(PFNGLBUFFERDATAPROC, glBufferData); LOAD(PFNGLBINDBUFFERBASEPROC, glBindBufferBase);
    LOAD(PFNGLGETBUFFERSUBDATAPROC, glGetBufferSubData); LOAD(PFNGLDELETEBUFFERSPROC, glDeleteBuffers); LOAD(PFNGLDISPATCHCOMPUTEPROC, glDispatchCompute);
    LOAD(PFNGLMEMORYBARRIERPROC, glMemoryBarrier); LOAD(PFNGLFINISHPROC, glFinish); LOAD(PFNGLGENVERTEXARRAYSPROC, glGenVertexArrays);
    LOAD(PFNGLBINDVERTEXARRAYPROC, glBindVertexArray); LOAD(PFNGLDELETEVERTEXARRAYSPROC, glDeleteVertexArrays); LOAD(PFNGLGETATTRIBLOCATIONPROC, glGetAttribLocation);
    LOAD(PFNGLENABLEVERTEXATTRIBARRAYPROC, glEnableVertexAttribArray); LOAD(PFNGLVERTEXATTRIBPOINTERPROC, glVertexAttribPointer);
    LOAD(PFNGLGENTEXTURESPROC, glGenTextures); LOAD(PFNGLBINDTEXTUREPROC, glBindTexture); LOAD(PFNGLTEXIMAGE2DPROC, glTexImage2D);
    LOAD(PFNGLTEXPARAMETERIPROC, glTexParameteri); LOAD(PFNGLDELETETEXTURESPROC, glDeleteTextures); LOAD(PFNGLGENFRAMEBUFFERSPROC, glGenFramebuffers);
    LOAD(PFNGLBINDFRAMEBUFFERPROC, glBindFramebuffer); LOAD(PFNGLFRAMEBUFFERTEXTURE2DPROC, glFramebufferTexture2D);
    LOAD(PFNGLCHECKFRAMEBUFFERSTATUSPROC, glCheckFramebufferStatus); LOAD(PFNGLDELETEFRAMEBUFFERSPROC, glDeleteFramebuffers);
    LOAD(PFNGLVIEWPORTPROC, glViewport); LOAD(PFNGLENABLEPROC, glEnable); LOAD(PFNGLDISABLEPROC, glDisable); LOAD(PFNGLBLENDEQUATIONPROC, glBlendEquation);
    LOAD(PFNGLBLENDFUNCPROC, glBlendFunc); LOAD(PFNGLCLEARCOLORPROC, glClearColor); LOAD(PFNGLCLEARPROC, glClear); LOAD(PFNGLDRAWELEMENTSPROC, glDrawElements);
    LOAD(PFNGLREADPIXELSPROC, glReadPixels); LOAD(PFNGLPIXELSTOREIPROC, glPixelStorei);
    LOAD(PFNGLGENRENDERBUFFERSPROC, glGenRenderbuffers); LOAD(PFNGLBINDRENDERBUFFERPROC, glBindRenderbuffer);
    LOAD(PFNGLRENDERBUFFERSTORAGEPROC, glRenderbufferStorage); LOAD(PFNGLFRAMEBUFFERRENDERBUFFERPROC, glFramebufferRenderbuffer);
    LOAD(PFNGLDELETERENDERBUFFERSPROC, glDeleteRenderbuffers);
    LOAD(PFNGLGENERATEMIPMAPPROC, glGenerateMipmap); LOAD(PFNGLACTIVETEXTUREPROC, glActiveTexture); LOAD(PFNGLUNIFORM1IPROC, glUniform1i);
    LOAD(PFNGLUNIFORM1FPROC, glUniform1f); LOAD(PFNGLGETTEXIMAGEPROC, glGetTexImage);
#undef LOAD
    snprintf(g_version, sizeof(g_version), "%s / %s / GLSL %s", (const char*)pglGetString(GL_VERSION), (const char*)pglGetString(GL_RENDERER),
             (const char*)pglGetString(GL_SHADING_LANGUAGE_VERSION));
    GLuint cs = compile(GL_COMPUTE_SHADER, shader_dir, "presort_compute.glsl", "");      /* splatrenderer.cpp:79-84 */
    if (!cs) return -1;
    g_presort = link_program(cs, 0, 0);
    if (!g_presort) return -1;
    g_ready = 1;
    return glref_configure(full_sh, srgb);
}

/* SplatRenderer::Init: the splat program with its defines (splatrenderer.cpp:59-77) */
static int glref_configure(int full_sh, int srgb)
{
    full_sh = full_sh != 0; srgb = srgb != 0;
    if (!g_splat_cfg[full_sh][srgb]) {
        char defines[128] = "";
        if (full_sh) strcat(defines, "#define FULL_SH\n");
        if (srgb) strcat(defines, "#define FRAMEBUFFER_SRGB\n");
        GLuint vs = compile(GL_VERTEX_SHADER, g_dir, "splat_vert.glsl", defines);
        GLuint gs = vs ? compile(GL_GEOMETRY_SHADER, g_dir, "splat_geom.glsl", defines) : 0;
        GLuint fs = gs ? compile(GL_FRAGMENT_SHADER, g_dir, "splat_frag.glsl", defines) : 0;
        if (!fs) return -1;
        g_splat_cfg[full_sh][srgb] = link_program(vs, gs, fs);
        if (!g_splat_cfg[full_sh][srgb]) return -1;
    }
    g_splat = g_splat_cfg[full_sh][srgb];
    g_full_sh = full_sh;
    return 0;
}

const char* glref_gl_version(void) { return g_version; }

/* SplatRenderer::Sort's pre-sort (splatrenderer.cpp:171-204): pos4 = the posVec of :106-111, mvp = projMat * inverse(cameraMat)
 * (computed by the caller: glm's arithmetic is not part of this harness), keyMax = 0xFFFFFFFF.  keys_out / idx_out receive the
 * *count_out entries in the order the atomic counter handed out slots (not deterministic, :50 of the shader). */
int glref_presort(const float* pos4, uint32_t n, const float mvp[16], const float nearFar[2], uint32_t* keys_out, uint32_t* idx_out,
                  uint32_t* count_out)
{
    if (!g_ready) return fail("glref_init has not run");
    GLuint b[4];
    pglGenBuffers(4, b);
    pglBindBuffer(GL_SHADER_STORAGE_BUFFER, b[0]);
    pglBufferData(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)n * 16, pos4, GL_STATIC_DRAW);
    pglBindBuffer(GL_SHADER_STORAGE_BUFFER, b[1]);
    pglBufferData(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)n * 4, NULL, GL_DYNAMIC_READ);
    pglBindBuffer(GL_SHADER_STORAGE_BUFFER, b[2]);
    pglBufferData(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)n * 4, NULL, GL_DYNAMIC_READ);
    const uint32_t zero = 0;
    pglBindBuffer(GL_ATOMIC_COUNTER_BUFFER, b[3]);
    pglBufferData(GL_ATOMIC_COUNTER_BUFFER, 4, &zero, GL_DYNAMIC_READ);
    pglUseProgram(g_presort);
    pglUniformMatrix4fv(pglGetUniformLocation(g_presort, "modelViewProj"), 1, GL_FALSE, mvp);
    pglUniform2fv(pglGetUniformLocation(g_presort, "nearFar"), 1, nearFar);
    pglUniform1ui(pglGetUniformLocation(g_presort, "keyMax"), 0xFFFFFFFFu);
    pglBindBufferBase(GL_SHADER_STORAGE_BUFFER, 0, b[0]);
    pglBindBufferBase(GL_SHADER_STORAGE_BUFFER, 1, b[1]);
    pglBindBufferBase(GL_SHADER_STORAGE_BUFFER, 2, b[2]);
    pglBindBufferBase(GL_ATOMIC_COUNTER_BUFFER, 4, b[3]);
    pglDispatchCompute((n + 255u) / 256u, 1, 1);
    pglMemoryBarrier(GL_SHADER_STORAGE_BARRIER_BIT | GL_ATOMIC_COUNTER_BARRIER_BIT | GL_BUFFER_UPDATE_BARRIER_BIT);
    pglFinish();
    uint32_t cnt = 0;
    pglBindBuffer(GL_ATOMIC_COUNTER_BUFFER, b[3]);
    pglGetBufferSubData(GL_ATOMIC_COUNTER_BUFFER, 0, 4, &cnt);
    if (cnt > n) cnt = n;
    pglBindBuffer(GL_SHADER_STORAGE_BUFFER, b[1]);
    pglGetBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, (GLsizeiptr)cnt * 4, keys_out);
    pglBindBuffer(GL_SHADER_STORAGE_BUFFER, b[2]);
    pglGetBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, (GLsizeiptr)cnt * 4, idx_out);
    *count_out = cnt;
    pglDeleteBuffers(4, b);
    const GLenum e = pglGetError();
    return e == GL_NO_ERROR ? 0 : fail("GL error 0x%x in glref_presort", e);
}

/* SplatRenderer::Render (splatrenderer.cpp:315-343) with App's Clear() state (app.cpp:144-164).  aos: the GaussianCloud records
 * (25 or 61 floats per splat, gaussiancloud.cpp:32-56); sorted_idx[count]: the element buffer (draw order).  rgba_out: H x W x 4
 * floats, row 0 = the GL bottom row.
 * target: 0 = RGBA32F, the colour-only target of --fp32 (app.cpp:1018-1027); 1 = RGBA8 like the default back buffer (the ROP
 * clamps and stores 8-bit unorm after every blend); 2 = RGBA16F, the --fp16 target (app.cpp:1012-1014).
 * depth_bits: 0 = colour-only (GL_DEPTH_TEST inert); 24 / 32 = a depth attachment like the default back buffer's (sdl_main.cpp:79)
 * or a float one: GL_DEPTH_TEST, enabled by Clear(), is then live (GL_LESS, depth writes on, cleared to 1). */
int glref_render_target(const float* aos, uint32_t n, const uint32_t* sorted_idx, uint32_t count, const float viewMat[16],
                        const float projMat[16], const float viewport[4], const float nearFar[2], const float eye[3], int target,
                        int depth_bits, float* rgba_out)
{
    if (!g_ready) return fail("glref_init has not run");
    const int W = (int)viewport[2], H = (int)viewport[3];
    const int stride = (g_full_sh ? 61 : 25) * 4;
    GLuint vao, vbo, ebo, tex, fbo, rbo = 0;
    pglGenVertexArrays(1, &vao);
    pglBindVertexArray(vao);
    pglGenBuffers(1, &vbo);
    pglBindBuffer(GL_ARRAY_BUFFER, vbo);
    pglBufferData(GL_ARRAY_BUFFER, (GLsizeiptr)n * stride, aos, GL_STATIC_DRAW);
    pglGenBuffers(1, &ebo);
    pglBindBuffer(GL_ELEMENT_ARRAY_BUFFER, ebo);
    pglBufferData(GL_ELEMENT_ARRAY_BUFFER, (GLsizeiptr)count * 4, sorted_idx, GL_STATIC_DRAW);
    /* BuildVertexArrayObject (splatrenderer.cpp:345-391): attribute NAME -> (components, float offset) of the interleaved record */
    struct attr { const char* name; int comps, off; };
    const struct attr base[] = {{"position", 4, 0}, {"r_sh0", 4, 4}, {"g_sh0", 4, 8}, {"b_sh0", 4, 12},
                                {"cov3_col0", 3, 16}, {"cov3_col1", 3, 19}, {"cov3_col2", 3, 22}};
    const struct attr full[] = {{"r_sh1", 4, 25}, {"r_sh2", 4, 29}, {"r_sh3", 4, 33}, {"g_sh1", 4, 37}, {"g_sh2", 4, 41},
                                {"g_sh3", 4, 45}, {"b_sh1", 4, 49}, {"b_sh2", 4, 53}, {"b_sh3", 4, 57}};
    for (int pass = 0; pass < (g_full_sh ? 2 : 1); ++pass) {
        const int cnt = pass ? 9 : 7;
        for (int k = 0; k < cnt; ++k) {
            const struct attr* a = pass ? &full[k] : &base[k];
            const GLint loc = pglGetAttribLocation(g_splat, a->name);
            if (loc < 0) return fail("attribute %s not active in the splat program", a->name);
            pglEnableVertexAttribArray((GLuint)loc);
            pglVertexAttribPointer((GLuint)loc, a->comps, GL_FLOAT, GL_FALSE, stride, (const void*)(size_t)(a->off * 4));
        }
    }
    pglGenTextures(1, &tex);
    pglBindTexture(GL_TEXTURE_2D, tex);
    pglTexImage2D(GL_TEXTURE_2D, 0, target == 1 ? GL_RGBA8 : (target == 2 ? GL_RGBA16F : GL_RGBA32F), W, H, 0, GL_RGBA, GL_FLOAT, NULL);
    pglTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    pglTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    pglGenFramebuffers(1, &fbo);
    pglBindFramebuffer(GL_FRAMEBUFFER, fbo);
    pglFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, tex, 0);
    if (depth_bits) {
        pglGenRenderbuffers(1, &rbo);
        pglBindRenderbuffer(GL_RENDERBUFFER, rbo);
        pglRenderbufferStorage(GL_RENDERBUFFER, depth_bits == 32 ? GL_DEPTH_COMPONENT32F : GL_DEPTH_COMPONENT24, W, H);
        pglFramebufferRenderbuffer(GL_FRAMEBUFFER, GL_DEPTH_ATTACHMENT, GL_RENDERBUFFER, rbo);
    }
    if (pglCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) return fail("framebuffer incomplete");
    /* Clear() of app.cpp:144-164 */
    pglViewport(0, 0, W, H);
    pglEnable(GL_BLEND);
    pglBlendEquation(GL_FUNC_ADD);
    pglBlendFunc(GL_ONE, GL_ONE_MINUS_SRC_ALPHA);
    pglClearColor(0.0f, 0.0f, 0.0f, 1.0f);
    pglClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT);
    pglEnable(GL_DEPTH_TEST);                     /* inert while the target is colour-only (app.cpp:1027) */
    /* Render's uniforms (splatrenderer.cpp:327-335) */
    pglUseProgram(g_splat);
    pglUniformMatrix4fv(pglGetUniformLocation(g_splat, "viewMat"), 1, GL_FALSE, viewMat);
    pglUniformMatrix4fv(pglGetUniformLocation(g_splat, "projMat"), 1, GL_FALSE, projMat);
    pglUniform4fv(pglGetUniformLocation(g_splat, "viewport"), 1, viewport);
    const float projParams[4] = {0.0f, nearFar[0], nearFar[1], 0.0f};
    pglUniform4fv(pglGetUniformLocation(g_splat, "projParams"), 1, projParams);
    pglUniform3fv(pglGetUniformLocation(g_splat, "eye"), 1, eye);
    pglDrawElements(GL_POINTS, (GLsizei)count, GL_UNSIGNED_INT, NULL);
    pglFinish();
    pglPixelStorei(GL_PACK_ALIGNMENT, 1);
    pglReadPixels(0, 0, W, H, GL_RGBA, GL_FLOAT, rgba_out);
    const GLenum e = pglGetError();
    pglBindFramebuffer(GL_FRAMEBUFFER, 0);
    pglDeleteFramebuffers(1, &fbo);
    if (rbo) pglDeleteRenderbuffers(1, &rbo);
    pglDeleteTextures(1, &tex);
    pglBindVertexArray(0);
    pglDeleteVertexArrays(1, &vao);
    pglDeleteBuffers(1, &vbo);
    pglDeleteBuffers(1, &ebo);
    return e == GL_NO_ERROR ? 0 : fail("GL error 0x%x in glref_render", e);
}

int glref_render(const float* aos, uint32_t n, const uint32_t* sorted_idx, uint32_t count, const float viewMat[16], const float projMat[16],
                 const float viewport[4], const float nearFar[2], const float eye[3], float* rgba_out)
{
    return glref_render_target(aos, n, sorted_idx, count, viewMat, projMat, viewport, nearFar, eye, 0, 0, rgba_out);
}

/* PointRenderer::Render's draw (pointrenderer.cpp:168-195) on the reference's point_vert / point_geom / point_frag, with the texture
 * PointRenderer::Init makes of the sprite (pointrenderer.cpp:54-63, core/texture.cpp:47-77: GL_RGBA8 or GL_SRGB8_ALPHA8 from 8-bit
 * RGBA, glGenerateMipmap, LinearMipmapLinear / Linear, ClampToEdge) and App's Clear() state.
 * points: n x 8 floats (position.xyzw, color.rgba -- PointCloud's interleaved record, BuildVertexArrayObject :198-226);
 * sprite_rgba8: tw x th texels AS UPLOADED, i.e. after Image::Load's row flip and 8-bit alpha pre-multiplication (the caller
 * restates that host code); srgb_tex: Image::isSRGB (= isFramebufferSRGBEnabled, :60).  depth_bits as in glref_render_target.
 * mips_out (optional): the float RGBA texels of every mip level the GL made, level after level (what glGenerateMipmap produced). */
int glref_points_render(const float* points, uint32_t n, const uint32_t* sorted_idx, uint32_t count, const float modelViewMat[16],
                        const float projMat[16], const float viewport[4], const uint8_t* sprite_rgba8, int tw, int th, int srgb_tex,
                        int depth_bits, float* rgba_out, float* mips_out)
{
    if (!g_ready) return fail("glref_init has not run");
    if (!g_point) {
        GLuint vs = compile(GL_VERTEX_SHADER, g_dir, "point_vert.glsl", "");
        GLuint gs = vs ? compile(GL_GEOMETRY_SHADER, g_dir, "point_geom.glsl", "") : 0;
        GLuint fs = gs ? compile(GL_FRAGMENT_SHADER, g_dir, "point_frag.glsl", "") : 0;
        if (!fs) return -1;
        g_point = link_program(vs, gs, fs);
        if (!g_point) return -1;
    }
    const int W = (int)viewport[2], H = (int)viewport[3];
    GLuint vao, vbo, ebo, tex, fbo, rbo = 0, sprite;
    pglGenVertexArrays(1, &vao);
    pglBindVertexArray(vao);
    pglGenBuffers(1, &vbo);
    pglBindBuffer(GL_ARRAY_BUFFER, vbo);
    pglBufferData(GL_ARRAY_BUFFER, (GLsizeiptr)n * 32, points, GL_STATIC_DRAW);
    pglGenBuffers(1, &ebo);
    pglBindBuffer(GL_ELEMENT_ARRAY_BUFFER, ebo);
    pglBufferData(GL_ELEMENT_ARRAY_BUFFER, (GLsizeiptr)count * 4, sorted_idx, GL_STATIC_DRAW);
    const char* names[2] = {"position", "color"};
    for (int k = 0; k < 2; ++k) {
        const GLint loc = pglGetAttribLocation(g_point, names[k]);
        if (loc < 0) return fail("attribute %s not active in the point program", names[k]);
        pglEnableVertexAttribArray((GLuint)loc);
        pglVertexAttribPointer((GLuint)loc, 4, GL_FLOAT, GL_FALSE, 32, (const void*)(size_t)(k * 16));
    }
    /* Texture::Texture(image, params), core/texture.cpp:47-77 */
    pglGenTextures(1, &sprite);
    pglBindTexture(GL_TEXTURE_2D, sprite);
    pglTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR_MIPMAP_LINEAR);
    pglTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
    pglTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    pglTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
    pglPixelStorei(GL_UNPACK_ALIGNMENT, 1);
    pglTexImage2D(GL_TEXTURE_2D, 0, srgb_tex ? GL_SRGB8_ALPHA8 : GL_RGBA, tw, th, 0, GL_RGBA, GL_UNSIGNED_BYTE, sprite_rgba8);
    pglGenerateMipmap(GL_TEXTURE_2D);
    if (mips_out) {
        float* o = mips_out;
        pglPixelStorei(GL_PACK_ALIGNMENT, 1);
        for (int l = 0, w = tw, h = th;; ++l) {
            pglGetTexImage(GL_TEXTURE_2D, l, GL_RGBA, GL_FLOAT, o);
            o += (size_t)w * h * 4;
            if (w == 1 && h == 1) break;
            w = w > 1 ? w / 2 : 1;
            h = h > 1 ? h / 2 : 1;
        }
    }
    pglGenTextures(1, &tex);
    pglBindTexture(GL_TEXTURE_2D, tex);
    pglTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, W, H, 0, GL_RGBA, GL_FLOAT, NULL);
    pglTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    pglTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    pglGenFramebuffers(1, &fbo);
    pglBindFramebuffer(GL_FRAMEBUFFER, fbo);
    pglFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, tex, 0);
    if (depth_bits) {
        pglGenRenderbuffers(1, &rbo);
        pglBindRenderbuffer(GL_RENDERBUFFER, rbo);
        pglRenderbufferStorage(GL_RENDERBUFFER, depth_bits == 32 ? GL_DEPTH_COMPONENT32F : GL_DEPTH_COMPONENT24, W, H);
        pglFramebufferRenderbuffer(GL_FRAMEBUFFER, GL_DEPTH_ATTACHMENT, GL_RENDERBUFFER, rbo);
    }
    if (pglCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) return fail("framebuffer incomplete");
    pglViewport(0, 0, W, H);
    pglEnable(GL_BLEND);
    pglBlendEquation(GL_FUNC_ADD);
    pglBlendFunc(GL_ONE, GL_ONE_MINUS_SRC_ALPHA);
    pglClearColor(0.0f, 0.0f, 0.0f, 1.0f);
    pglClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT);
    pglEnable(GL_DEPTH_TEST);
    pglUseProgram(g_point);
    pglUniformMatrix4fv(pglGetUniformLocation(g_point, "modelViewMat"), 1, GL_FALSE, modelViewMat);
    pglUniformMatrix4fv(pglGetUniformLocation(g_point, "projMat"), 1, GL_FALSE, projMat);
    pglUniform1f(pglGetUniformLocation(g_point, "pointSize"), 0.02f);                              /* pointrenderer.cpp:182 */
    pglUniform1f(pglGetUniformLocation(g_point, "invAspectRatio"), 1.0f / (viewport[2] / viewport[3]));   /* :176-183 */
    pglActiveTexture(GL_TEXTURE0);
    pglBindTexture(GL_TEXTURE_2D, sprite);
    pglUniform1i(pglGetUniformLocation(g_point, "colorTex"), 0);
    pglDrawElements(GL_POINTS, (GLsizei)count, GL_UNSIGNED_INT, NULL);
    pglFinish();
    pglPixelStorei(GL_PACK_ALIGNMENT, 1);
    pglReadPixels(0, 0, W, H, GL_RGBA, GL_FLOAT, rgba_out);
    const GLenum e = pglGetError();
    pglBindFramebuffer(GL_FRAMEBUFFER, 0);
    pglDeleteFramebuffers(1, &fbo);
    if (rbo) pglDeleteRenderbuffers(1, &rbo);
    pglDeleteTextures(1, &tex);
    pglDeleteTextures(1, &sprite);
    pglBindVertexArray(0);
    pglDeleteVertexArrays(1, &vao);
    pglDeleteBuffers(1, &vbo);
    pglDeleteBuffers(1, &ebo);
    return e == GL_NO_ERROR ? 0 : fail("GL error 0x%x in glref_points_render", e);
}

/* ------------------------------------------------------------------------------------------------------------------------------
 * The reference's fallback sorter, rgc::radix_sort::sorter (src/radix_sort.hpp: the path SplatRenderer::Sort takes where
 * KHR_shader_subgroup is missing, splatrenderer.cpp:86,223-264 -- llvmpipe is such a GL).  Its three compute shaders are C string
 * literals inside that header (radix_sort.hpp:122-124); they are READ from the file where it lies at run time (nothing is copied),
 * unescaped, prefixed with "#version 460" like __rgc_shader_injector_load_src does (:93-103), and driven with the buffers,
 * uniforms and dispatch sequence of sorter::sort (:340-485): eight 4-bit passes of count -> Blelloch scan of the per-block
 * counts (up-sweep, clear last, down-sweep) -> reorder, ping-ponging between the caller's buffers and scratch.
 * keys / vals (n each) are sorted in place.
 * ------------------------------------------------------------------------------------------------------------------------------ */
static GLuint g_rgc[3];            /* count, local offsets, reorder */

/* the next string literal assigned to a variable whose name starts with `prefix`, unescaped (malloc'd); *cursor advances */
static char* next_literal(const char** cursor, const char* prefix)
{
    const char* p = strstr(*cursor, prefix);
    if (!p) return NULL;
    p = strchr(p, '"');
    if (!p) return NULL;
    ++p;
    size_t cap = 1 << 16, len = 0;
    char* out = (char*)malloc(cap);
    while (*p && *p != '"') {
        char c = *p++;
        if (c == '\\' && *p) {
            const char e = *p++;
            c = e == 'n' ? '\n' : e == 't' ? '\t' : e == 'r' ? '\r' : e == '0' ? '\0' : e;      /* \" \\ \' map to themselves */
        }
        if (len + 2 > cap) { cap *= 2; out = (char*)realloc(out, cap); }
        out[len++] = c;
    }
    out[len] = 0;
    *cursor = *p ? p + 1 : p;
    return out;
}

static GLuint compute_program_from_source(const char* body, const char* what)
{
    const size_t n = strlen(body) + 32;
    char* src = (char*)malloc(n);
    snprintf(src, n, "#version 460\n%s", body);                         /* radix_sort.hpp:93-103 (non-Android) */
    GLuint sh = pglCreateShader(GL_COMPUTE_SHADER);
    const char* srcs[1] = {src};
    pglShaderSource(sh, 1, srcs, NULL);
    pglCompileShader(sh);
    GLint ok = 0;
    pglGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
    free(src);
    if (!ok) {
        char log[3000];
        pglGetShaderInfoLog(sh, sizeof(log), NULL, log);
        fail("compile rgc %s shader: %s", what, log);
        return 0;
    }
    return link_program(sh, 0, 0);
}

static int rgc_load(const char* hpp_path)
{
    if (g_rgc[0]) return 0;
    FILE* f = fopen(hpp_path, "rb");
    if (!f) return fail("cannot open %s", hpp_path);
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* text = (char*)malloc((size_t)sz + 1);
    if (fread(text, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(text); return fail("short read of %s", hpp_path); }
    text[sz] = 0;
    fclose(f);
    const char* cur = text;
    char* lit;
    /* which literal is which program follows from the interface it declares (sorter::sorter attaches them by hash, :300-330) */
    while ((lit = next_literal(&cur, "__rgc_shader_injector_shader_src_")) != NULL) {
        int slot = -1;
        if (strstr(lit, "layout(local_size_x")) {
            if (strstr(lit, "b_tot_count_buf")) slot = 0;
            else if (strstr(lit, "u_op")) slot = 1;
            else if (strstr(lit, "b_out_keys")) slot = 2;
        }
        if (slot >= 0 && !g_rgc[slot]) {
            g_rgc[slot] = compute_program_from_source(lit, slot == 0 ? "count" : slot == 1 ? "local offsets" : "reorder");
            if (!g_rgc[slot]) { free(lit); free(text); return -1; }
        }
        free(lit);
    }
    free(text);
    if (!g_rgc[0] || !g_rgc[1] || !g_rgc[2]) return fail("%s: the sorter's three shader literals were not found", hpp_path);
    return 0;
}

int glref_rgc_sort(const char* hpp_path, uint32_t* keys, uint32_t* vals, uint32_t n)
{
    if (!g_ready) return fail("glref_init has not run");
    if (rgc_load(hpp_path)) return -1;
    if (n <= 1) return 0;                                                        /* sorter::sort, :342-344 */
    const uint32_t T = 64, I = 4, RADIX = 16;                                    /* RGC_RADIX_SORT_* (:80-84) */
    const uint32_t blocks = (uint32_t)ceilf((float)n / (float)(T * I));          /* calc_thread_blocks_num (:261-264) */
    const uint32_t blocks2 = (uint32_t)exp2(ceil(log2((double)blocks)));         /* round_to_power_of_2 (:266-270) */
    const size_t off_bytes = (size_t)blocks2 * RADIX * 4;
    uint32_t* zeros = (uint32_t*)calloc(off_bytes / 4 + RADIX, 4);
    GLuint b[6];                       /* keys, values, key scratch, value scratch, local offsets, global counts (resize_internal_buf, :272-300) */
    pglGenBuffers(6, b);
    for (int k = 0; k < 4; ++k) {
        pglBindBuffer(GL_SHADER_STORAGE_BUFFER, b[k]);
        pglBufferData(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)n * 4, k == 0 ? (const void*)keys : k == 1 ? (const void*)vals : NULL, GL_DYNAMIC_COPY);
    }
    const GLuint kb[2] = {b[0], b[2]}, vb[2] = {b[1], b[3]};
    const GLuint wg_scan = (GLuint)ceilf((float)blocks2 / (float)(T * I));
    const int depth = (int)log2((double)blocks2);
    for (uint32_t pass = 0; pass < 32 / 4; ++pass) {                            /* RGC_RADIX_SORT_BITSET_COUNT passes (:360) */
        /* initial clearing (:366-384) */
        pglBindBuffer(GL_SHADER_STORAGE_BUFFER, b[5]);
        pglBufferData(GL_SHADER_STORAGE_BUFFER, RADIX * 4, zeros, GL_DYNAMIC_COPY);
        pglBindBuffer(GL_SHADER_STORAGE_BUFFER, b[4]);
        pglBufferData(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)off_bytes, zeros, GL_DYNAMIC_COPY);
        /* counting (:390-404) */
        pglUseProgram(g_rgc[0]);
        pglBindBufferBase(GL_SHADER_STORAGE_BUFFER, 0, kb[pass % 2]);
        pglBindBufferBase(GL_SHADER_STORAGE_BUFFER, 1, b[4]);
        pglBindBufferBase(GL_SHADER_STORAGE_BUFFER, 2, b[5]);
        pglUniform1ui(pglGetUniformLocation(g_rgc[0], "u_arr_len"), n);
        pglUniform1ui(pglGetUniformLocation(g_rgc[0], "u_bitset_idx"), pass);
        pglDispatchCompute(blocks, 1, 1);
        pglMemoryBarrier(GL_SHADER_STORAGE_BARRIER_BIT);
        /* block-wide exclusive scan per radix (:410-452): up-sweep, clear last, down-sweep */
        pglUseProgram(g_rgc[1]);
        pglBindBufferBase(GL_SHADER_STORAGE_BUFFER, 0, b[4]);
        const GLint l_len = pglGetUniformLocation(g_rgc[1], "u_arr_len"), l_op = pglGetUniformLocation(g_rgc[1], "u_op"),
                    l_depth = pglGetUniformLocation(g_rgc[1], "u_depth");
        for (int d = 0; d < depth; ++d) {
            pglUniform1ui(l_len, blocks2); pglUniform1ui(l_op, 0); pglUniform1ui(l_depth, (GLuint)d);
            pglDispatchCompute(wg_scan, 1, 1);
            pglMemoryBarrier(GL_SHADER_STORAGE_BARRIER_BIT);
        }
        pglUniform1ui(l_len, blocks2); pglUniform1ui(l_op, 1);
        pglDispatchCompute(wg_scan, 1, 1);
        pglMemoryBarrier(GL_SHADER_STORAGE_BARRIER_BIT);
        for (int d = depth - 1; d >= 0; --d) {
            pglUniform1ui(l_len, blocks2); pglUniform1ui(l_op, 2); pglUniform1ui(l_depth, (GLuint)d);
            pglDispatchCompute(wg_scan, 1, 1);
            pglMemoryBarrier(GL_SHADER_STORAGE_BARRIER_BIT);
        }
        /* reordering (:458-480) */
        pglUseProgram(g_rgc[2]);
        pglBindBufferBase(GL_SHADER_STORAGE_BUFFER, 0, kb[pass % 2]);
        pglBindBufferBase(GL_SHADER_STORAGE_BUFFER, 1, kb[(pass + 1) % 2]);
        pglBindBufferBase(GL_SHADER_STORAGE_BUFFER, 2, vb[pass % 2]);
        pglBindBufferBase(GL_SHADER_STORAGE_BUFFER, 3, vb[(pass + 1) % 2]);
        pglBindBufferBase(GL_SHADER_STORAGE_BUFFER, 4, b[4]);
        pglBindBufferBase(GL_SHADER_STORAGE_BUFFER, 5, b[5]);
        pglUniform1ui(pglGetUniformLocation(g_rgc[2], "u_write_values"), 1u);
        pglUniform1ui(pglGetUniformLocation(g_rgc[2], "u_arr_len"), n);
        pglUniform1ui(pglGetUniformLocation(g_rgc[2], "u_bitset_idx"), pass);
        pglDispatchCompute(blocks, 1, 1);
        pglMemoryBarrier(GL_SHADER_STORAGE_BARRIER_BIT);
    }
    pglFinish();
    /* eight passes: the result is back in the caller's buffers */
    pglBindBuffer(GL_SHADER_STORAGE_BUFFER, b[0]);
    pglGetBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, (GLsizeiptr)n * 4, keys);
    pglBindBuffer(GL_SHADER_STORAGE_BUFFER, b[1]);
    pglGetBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, (GLsizeiptr)n * 4, vals);
    pglDeleteBuffers(6, b);
    free(zeros);
    const GLenum e = pglGetError();
    return e == GL_NO_ERROR ? 0 : fail("GL error 0x%x in glref_rgc_sort", e);
}
