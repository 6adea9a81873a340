"""A real OpenGL implementation for the renderer rows (TEST INFRASTRUCTURE ONLY).

The reference renders image A with OpenGL (vispy_renderer.py:47-178 through vispy + PyOpenGL; offscreen_renderer.py through
pyrender).  Neither vispy nor PyOpenGL nor a GPU GL driver exists in the offline image -- but a Khronos-conformant SOFTWARE
implementation does: Google SwiftShader (OpenGL ES 3.0, EGL 1.4, headless), shipped inside the `kaleido` wheel
(.../kaleido/executable/bin/swiftshader/libEGL.so, libGLESv2.so).  This module
  * binds the ~45 EGL / GLES entry points needed with ctypes (`GL`), and
  * provides in-memory stand-ins for the THIN layers between the reference class and GL: `vispy.app.Canvas`, the handful of
    `vispy.gloo` objects / calls vispy_renderer.py uses (each mapped to the 1-3 GL calls vispy itself issues for it),
    `OpenGL.GL.glReadPixels`, and `plyfile.PlyData.read`,
so that the UNMODIFIED reference class `VispyRenderer` runs on a real GL implementation (oracle/make_gl_golden.py).
What is this repo's and not the reference's or the GL implementation's: (1) the gloo -> GL mapping below (documented per
call, after vispy's gloo/glir.py: uniforms uploaded with transpose = GL_FALSE, `Texture2D(shape=(H, W, 3))` = RGB8,
`RenderBuffer` without a format attached as depth = GL_DEPTH_COMPONENT16, `set_cull_face` = glCullFace only, `clear(color=True)`
= the current clear colour); (2) the mechanical GLSL 1.30 -> GLSL ES 3.00 header translation (`#version`, `attribute` / `varying`
-> `in` / `out`, a default `highp` precision): the shader BODIES are the reference's, compiled by SwiftShader.
Everything that the numpy restatement (oracle/raster_oracle.py) had to assume about GL -- fill rule, clipping, perspective-correct
interpolation, depth test and depth-buffer quantisation, float -> unorm8 conversion, read-back row order -- is here decided by GL."""
import ctypes as C
import glob
import os
import sys
import types

import numpy as np

_SEARCH = ["/usr/local/lib/python3.10/dist-packages/kaleido/executable/bin/swiftshader",
           "/opt/conda/lib/python3.9/site-packages/kaleido/executable/bin/swiftshader"]


def find_swiftshader():
    dirs = list(_SEARCH)
    try:
        import kaleido
        dirs.insert(0, os.path.join(os.path.dirname(kaleido.__file__), "executable", "bin", "swiftshader"))
    except Exception:   # noqa: BLE001
        pass
    dirs += glob.glob("/usr/local/lib/python3*/dist-packages/kaleido/executable/bin/swiftshader")
    for d in dirs:
        if os.path.isfile(os.path.join(d, "libEGL.so")) and os.path.isfile(os.path.join(d, "libGLESv2.so")):
            return d
    return None


def available():
    return find_swiftshader() is not None


# ---- constants -------------------------------------------------------------------------------------------------------------
EGL_OPENGL_ES_API, EGL_NONE, EGL_SURFACE_TYPE, EGL_PBUFFER_BIT = 0x30A0, 0x3038, 0x3033, 0x0001
EGL_RENDERABLE_TYPE, EGL_OPENGL_ES3_BIT, EGL_CONTEXT_CLIENT_VERSION = 0x3040, 0x0040, 0x3098
EGL_WIDTH, EGL_HEIGHT = 0x3057, 0x3056
GL_VERTEX_SHADER, GL_FRAGMENT_SHADER, GL_COMPILE_STATUS, GL_LINK_STATUS = 0x8B31, 0x8B30, 0x8B81, 0x8B82
GL_ARRAY_BUFFER, GL_ELEMENT_ARRAY_BUFFER, GL_STATIC_DRAW = 0x8892, 0x8893, 0x88E4
GL_FLOAT, GL_UNSIGNED_BYTE, GL_UNSIGNED_INT = 0x1406, 0x1401, 0x1405
GL_TRIANGLES, GL_DEPTH_TEST, GL_CULL_FACE, GL_BACK, GL_FRONT = 0x0004, 0x0B71, 0x0B44, 0x0405, 0x0404
GL_COLOR_BUFFER_BIT, GL_DEPTH_BUFFER_BIT = 0x4000, 0x0100
GL_FRAMEBUFFER, GL_RENDERBUFFER, GL_COLOR_ATTACHMENT0, GL_DEPTH_ATTACHMENT = 0x8D40, 0x8D41, 0x8CE0, 0x8D00
GL_FRAMEBUFFER_COMPLETE = 0x8CD5
GL_TEXTURE_2D, GL_RGB, GL_RGBA, GL_RGB8, GL_RGBA8 = 0x0DE1, 0x1907, 0x1908, 0x8051, 0x8058
GL_DEPTH_COMPONENT, GL_DEPTH_COMPONENT16, GL_DEPTH_COMPONENT24, GL_DEPTH_COMPONENT32F = 0x1902, 0x81A5, 0x81A6, 0x8CAC
GL_TEXTURE_MIN_FILTER, GL_TEXTURE_MAG_FILTER, GL_TEXTURE_WRAP_S, GL_TEXTURE_WRAP_T = 0x2801, 0x2800, 0x2802, 0x2803
GL_NEAREST, GL_LINEAR, GL_LINEAR_MIPMAP_LINEAR, GL_REPEAT, GL_CLAMP_TO_EDGE = 0x2600, 0x2601, 0x2703, 0x2901, 0x812F
GL_TEXTURE0, GL_PACK_ALIGNMENT, GL_UNPACK_ALIGNMENT = 0x84C0, 0x0D05, 0x0CF5
GL_VERSION, GL_RENDERER, GL_EXTENSIONS = 0x1F02, 0x1F01, 0x1F03


class GL:
    """One headless SwiftShader context (pbuffer), made current on construction; attribute access = the GLES function."""
    _inst = None

    @classmethod
    def get(cls):
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst

    def __init__(self):
        d = find_swiftshader()
        if d is None:
            raise RuntimeError("SwiftShader (kaleido) not found in this image")
        self.gles = C.CDLL(os.path.join(d, "libGLESv2.so"), mode=C.RTLD_GLOBAL)
        self.egl = C.CDLL(os.path.join(d, "libEGL.so"), mode=C.RTLD_GLOBAL)
        e = self.egl
        e.eglGetDisplay.restype = C.c_void_p; e.eglGetDisplay.argtypes = [C.c_void_p]
        e.eglInitialize.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        e.eglChooseConfig.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
        e.eglCreatePbufferSurface.restype = C.c_void_p
        e.eglCreatePbufferSurface.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        e.eglCreateContext.restype = C.c_void_p
        e.eglCreateContext.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        e.eglMakeCurrent.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        e.eglBindAPI.argtypes = [C.c_uint]
        self.dpy = e.eglGetDisplay(None)
        major, minor = C.c_int(), C.c_int()
        assert e.eglInitialize(self.dpy, C.byref(major), C.byref(minor)), "eglInitialize failed"
        assert e.eglBindAPI(EGL_OPENGL_ES_API)
        attr = (C.c_int * 5)(EGL_SURFACE_TYPE, EGL_PBUFFER_BIT, EGL_RENDERABLE_TYPE, EGL_OPENGL_ES3_BIT, EGL_NONE)
        cfg, ncfg = C.c_void_p(), C.c_int()
        assert e.eglChooseConfig(self.dpy, attr, C.byref(cfg), 1, C.byref(ncfg)) and ncfg.value >= 1, "no EGL config"
        pattr = (C.c_int * 5)(EGL_WIDTH, 16, EGL_HEIGHT, 16, EGL_NONE)
        self.surf = e.eglCreatePbufferSurface(self.dpy, cfg, pattr)
        cattr = (C.c_int * 3)(EGL_CONTEXT_CLIENT_VERSION, 3, EGL_NONE)
        self.ctx = e.eglCreateContext(self.dpy, cfg, None, cattr)
        assert self.ctx, "eglCreateContext failed"
        assert e.eglMakeCurrent(self.dpy, self.surf, self.surf, self.ctx), "eglMakeCurrent failed"
        g = self.gles
        g.glGetString.restype = C.c_char_p; g.glGetString.argtypes = [C.c_uint]
        g.glCreateShader.restype = C.c_uint; g.glCreateProgram.restype = C.c_uint
        g.glGetAttribLocation.restype = C.c_int; g.glGetAttribLocation.argtypes = [C.c_uint, C.c_char_p]
        g.glGetUniformLocation.restype = C.c_int; g.glGetUniformLocation.argtypes = [C.c_uint, C.c_char_p]
        g.glVertexAttribPointer.argtypes = [C.c_uint, C.c_int, C.c_uint, C.c_ubyte, C.c_int, C.c_void_p]
        g.glBufferData.argtypes = [C.c_uint, C.c_ssize_t, C.c_void_p, C.c_uint]
        g.glDrawElements.argtypes = [C.c_uint, C.c_int, C.c_uint, C.c_void_p]
        g.glReadPixels.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p]
        g.glTexImage2D.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_void_p]
        g.glClearColor.argtypes = [C.c_float] * 4
        g.glClearDepthf.argtypes = [C.c_float]
        g.glUniform3f.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float]
        g.glUniformMatrix4fv.argtypes = [C.c_int, C.c_int, C.c_ubyte, C.c_void_p]
        self.version = g.glGetString(GL_VERSION).decode()
        self.renderer = g.glGetString(GL_RENDERER).decode()
        self.extensions = g.glGetString(GL_EXTENSIONS).decode().split()

    def __getattr__(self, name):
        return getattr(self.gles, name)

    def check(self, what=""):
        err = self.gles.glGetError()
        if err:
            raise RuntimeError("GL error 0x%x after %s" % (err, what))


# ---- GLSL 1.30 -> GLSL ES 3.00 header translation (bodies untouched) ------------------------------------------------------------
def to_essl300(src, stage):
    out = []
    for line in src.splitlines():
        st = line.strip()
        if st.startswith("#version"):
            out.append("#version 300 es")
            out.append("precision highp float;")
            out.append("precision highp int;")
            continue
        if st.startswith("attribute "):
            line = line.replace("attribute ", "in ", 1)
        elif st.startswith("varying "):
            line = line.replace("varying ", "out " if stage == "vertex" else "in ", 1)
        out.append(line)
    # the #version directive must be the first line of the string: strip the indentation / blank lines in front of it
    txt = "\n".join(out)
    return txt[txt.index("#version"):]


# ---- vispy.gloo stand-ins: the calls vispy_renderer.py makes, mapped to GL as vispy's gloo / glir do ------------------------------
class _VertexBuffer:
    def __init__(self, data):
        gl = GL.get()
        self.data = np.ascontiguousarray(data)                    # structured array: interleaved attributes, as vispy uploads it
        self.id = C.c_uint()
        gl.glGenBuffers(1, C.byref(self.id))
        gl.glBindBuffer(GL_ARRAY_BUFFER, self.id)
        gl.glBufferData(GL_ARRAY_BUFFER, self.data.nbytes, self.data.ctypes.data, GL_STATIC_DRAW)
        gl.check("VertexBuffer")


class _IndexBuffer:
    def __init__(self, data):
        gl = GL.get()
        self.data = np.ascontiguousarray(data, np.uint32)
        self.id = C.c_uint()
        gl.glGenBuffers(1, C.byref(self.id))
        gl.glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, self.id)
        gl.glBufferData(GL_ELEMENT_ARRAY_BUFFER, self.data.nbytes, self.data.ctypes.data, GL_STATIC_DRAW)
        gl.check("IndexBuffer")


class _Program:
    def __init__(self, vert, frag):
        gl = GL.get()
        self.sources = (vert, frag)                               # the reference's strings, kept for inspection
        self.id = gl.glCreateProgram()
        for kind, src, stage in ((GL_VERTEX_SHADER, vert, "vertex"), (GL_FRAGMENT_SHADER, frag, "fragment")):
            sh = gl.glCreateShader(kind)
            txt = to_essl300(src, stage).encode()
            p = C.c_char_p(txt)
            gl.glShaderSource(sh, 1, C.byref(p), None)
            gl.glCompileShader(sh)
            ok = C.c_int()
            gl.glGetShaderiv(sh, GL_COMPILE_STATUS, C.byref(ok))
            if not ok.value:
                log = C.create_string_buffer(4096)
                gl.glGetShaderInfoLog(sh, 4096, None, log)
                raise RuntimeError("%s shader does not compile as GLSL ES 3.00: %s\n%s" % (stage, log.value.decode(), txt.decode()))
            gl.glAttachShader(self.id, sh)
        gl.glLinkProgram(self.id)
        ok = C.c_int()
        gl.glGetProgramiv(self.id, GL_LINK_STATUS, C.byref(ok))
        if not ok.value:
            log = C.create_string_buffer(4096)
            gl.glGetProgramInfoLog(self.id, 4096, None, log)
            raise RuntimeError("program link failed: " + log.value.decode())
        self.vbo = None

    def bind(self, vbo):                                           # gloo.Program.bind: one attribute per field of the structured array
        gl = GL.get()
        self.vbo = vbo
        self.vao = C.c_uint()
        gl.glGenVertexArrays(1, C.byref(self.vao))
        gl.glBindVertexArray(self.vao)
        gl.glBindBuffer(GL_ARRAY_BUFFER, vbo.id)
        dt = vbo.data.dtype
        for name in dt.names:
            loc = gl.glGetAttribLocation(self.id, name.encode())
            if loc < 0:
                continue                                           # (an attribute the compiler optimised away)
            sub, off = dt.fields[name][0], dt.fields[name][1]
            assert sub.base == np.float32
            gl.glEnableVertexAttribArray(loc)
            gl.glVertexAttribPointer(loc, int(np.prod(sub.shape)), GL_FLOAT, 0, dt.itemsize, C.c_void_p(off))
        gl.check("Program.bind")

    def __setitem__(self, name, value):                            # uniforms: glUniformMatrix4fv(transpose = GL_FALSE) / glUniform3f
        gl = GL.get()
        gl.glUseProgram(self.id)
        loc = gl.glGetUniformLocation(self.id, name.encode())
        v = np.ascontiguousarray(value, np.float32)
        if loc < 0:
            return
        if v.shape == (4, 4):
            gl.glUniformMatrix4fv(loc, 1, 0, v.ctypes.data)        # the array's C-order bytes as column-major data, as vispy does
        elif v.size == 3:
            gl.glUniform3f(loc, float(v.flat[0]), float(v.flat[1]), float(v.flat[2]))
        else:
            raise NotImplementedError(name)
        gl.check("uniform " + name)

    def draw(self, mode, indices):
        gl = GL.get()
        assert mode == "triangles"
        gl.glUseProgram(self.id)
        gl.glBindVertexArray(self.vao)
        gl.glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, indices.id)
        gl.glDrawElements(GL_TRIANGLES, int(indices.data.size), GL_UNSIGNED_INT, None)
        gl.check("draw")


class _Texture2D:
    def __init__(self, shape=None, data=None):
        gl = GL.get()
        h, w, c = shape
        assert c == 3
        self.shape = shape
        self.id = C.c_uint()
        gl.glGenTextures(1, C.byref(self.id))
        gl.glBindTexture(GL_TEXTURE_2D, self.id)
        gl.glTexImage2D(GL_TEXTURE_2D, 0, GL_RGB8, w, h, 0, GL_RGB, GL_UNSIGNED_BYTE, None)
        gl.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST)
        gl.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST)
        gl.check("Texture2D")


# vispy attaches a format-less RenderBuffer as GL_DEPTH_COMPONENT16 (gloo/glir.py: GlirFrameBuffer._formats).  SwiftShader cannot read a
# fixed-point depth buffer back correctly through GL_NV_read_depth (16 bit: the rows come back in its internal 2x2-quad order; 24 bit: 0 / 1
# only), so the stand-in attaches GL_DEPTH_COMPONENT32F: identical visibility for the test scenes (front / back surfaces are centimetres
# apart), read-back values within one 16-bit quantum (0.06-0.3 mm at 0.65-1.5 m) of what a 16-bit buffer would return.
DEPTH_FORMAT = {"bits": 32}


class _RenderBuffer:
    def __init__(self, shape=None, format=None):
        gl = GL.get()
        h, w = shape[:2]
        self.id = C.c_uint()
        gl.glGenRenderbuffers(1, C.byref(self.id))
        gl.glBindRenderbuffer(GL_RENDERBUFFER, self.id)
        fmt = {16: GL_DEPTH_COMPONENT16, 24: GL_DEPTH_COMPONENT24, 32: GL_DEPTH_COMPONENT32F}[DEPTH_FORMAT["bits"]]
        gl.glRenderbufferStorage(GL_RENDERBUFFER, fmt, w, h)
        gl.check("RenderBuffer")


class _FrameBuffer:
    def __init__(self, color=None, depth=None):
        gl = GL.get()
        self.id = C.c_uint()
        gl.glGenFramebuffers(1, C.byref(self.id))
        gl.glBindFramebuffer(GL_FRAMEBUFFER, self.id)
        gl.glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, color.id, 0)
        gl.glFramebufferRenderbuffer(GL_FRAMEBUFFER, GL_DEPTH_ATTACHMENT, GL_RENDERBUFFER, depth.id)
        st = gl.glCheckFramebufferStatus(GL_FRAMEBUFFER)
        assert st == GL_FRAMEBUFFER_COMPLETE, hex(st)
        gl.glBindFramebuffer(GL_FRAMEBUFFER, 0)

    def __enter__(self):
        GL.get().glBindFramebuffer(GL_FRAMEBUFFER, self.id)
        return self

    def __exit__(self, *a):
        GL.get().glBindFramebuffer(GL_FRAMEBUFFER, 0)


def _set_state(depth_test=None, **kw):
    assert not kw
    if depth_test:
        GL.get().glEnable(GL_DEPTH_TEST)


def _set_cull_face(mode="back"):
    GL.get().glCullFace({"back": GL_BACK, "front": GL_FRONT}[mode])   # selects the face; GL_CULL_FACE itself stays disabled


def _clear(color=True, depth=True):
    assert color is True and depth is True                            # "True" = clear with the current clear values
    GL.get().glClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT)


def _set_viewport(x, y, w, h):
    GL.get().glViewport(int(x), int(y), int(w), int(h))


def _glReadPixels(x, y, w, h, fmt, typ):
    """PyOpenGL's glReadPixels for the two calls of vispy_renderer.py:160-161.  GL ES guarantees RGBA / UNSIGNED_BYTE for a
    unorm colour buffer (RGB is implementation-defined), so the colour is read as RGBA and the alpha dropped; the depth buffer
    is read as GL_DEPTH_COMPONENT / GL_FLOAT through GL_NV_read_depth exactly as the reference asks."""
    gl = GL.get()
    gl.glPixelStorei(GL_PACK_ALIGNMENT, 1)
    gl.glFinish()
    if fmt == GL_RGB and typ == GL_UNSIGNED_BYTE:
        buf = np.zeros((h, w, 4), np.uint8)
        gl.glReadPixels(x, y, w, h, GL_RGBA, GL_UNSIGNED_BYTE, buf.ctypes.data)
        gl.check("glReadPixels RGBA")
        return np.ascontiguousarray(buf[..., :3]).tobytes()
    if fmt == GL_DEPTH_COMPONENT and typ == GL_FLOAT:
        assert "GL_NV_read_depth" in gl.extensions
        assert DEPTH_FORMAT["bits"] == 32, "only the float depth attachment reads back correctly on SwiftShader (see DEPTH_FORMAT)"
        buf = np.zeros((h, w), np.float32)
        gl.glReadPixels(x, y, w, h, GL_DEPTH_COMPONENT, GL_FLOAT, buf.ctypes.data)
        gl.check("glReadPixels DEPTH")
        return buf
    raise NotImplementedError((fmt, typ))


class _Canvas:
    def __init__(self, show=False, size=None, **kw):
        GL.get()
        self._size = size

    def update(self):
        pass


class _PlyData:
    """plyfile.PlyData.read for the ascii / binary PLYs the fixtures write: ply['vertex']['x'], ply['face']['vertex_indices']."""

    def __init__(self, elems):
        self._e = elems

    def __getitem__(self, k):
        return self._e[k]

    @staticmethod
    def read(path):
        from . import ply_io
        raw = ply_io.read_ply(path)
        out = {"vertex": {k: np.asarray(v) for k, v in raw["vertex"].items()}}
        f = raw["face"]["vertex_indices"]
        out["face"] = {"vertex_indices": [np.asarray(r) for r in np.asarray(f)]}
        return _PlyData(out)


def install_stubs():
    """Put the stand-ins into sys.modules (vispy, vispy.app, vispy.gloo, OpenGL, OpenGL.GL, plyfile)."""
    vispy = types.ModuleType("vispy")
    app = types.ModuleType("vispy.app")
    app.Canvas = _Canvas
    app.use_app = lambda backend=None: None
    gloo = types.ModuleType("vispy.gloo")
    gloo.Program, gloo.VertexBuffer, gloo.IndexBuffer = _Program, _VertexBuffer, _IndexBuffer
    gloo.FrameBuffer, gloo.Texture2D, gloo.RenderBuffer = _FrameBuffer, _Texture2D, _RenderBuffer
    gloo.set_state, gloo.set_cull_face, gloo.clear, gloo.set_viewport = _set_state, _set_cull_face, _clear, _set_viewport
    vispy.app, vispy.gloo = app, gloo
    ogl = types.ModuleType("OpenGL")
    oglgl = types.ModuleType("OpenGL.GL")
    oglgl.glReadPixels = _glReadPixels
    oglgl.GL_RGB, oglgl.GL_UNSIGNED_BYTE, oglgl.GL_DEPTH_COMPONENT, oglgl.GL_FLOAT = GL_RGB, GL_UNSIGNED_BYTE, GL_DEPTH_COMPONENT, GL_FLOAT
    ogl.GL = oglgl
    ply = types.ModuleType("plyfile")
    ply.PlyData, ply.PlyElement = _PlyData, object
    for name, mod in (("vispy", vispy), ("vispy.app", app), ("vispy.gloo", gloo), ("OpenGL", ogl), ("OpenGL.GL", oglgl), ("plyfile", ply)):
        sys.modules[name] = mod


# ---- the second renderer's GL work (offscreen_renderer.py:48-83 through pyrender), stated by this repo -------------------------------
# pyrender / trimesh cannot be installed offline, so the REFERENCE class cannot run here; what follows issues the GL calls pyrender
# makes for that scene -- IntrinsicsCamera projection, pose = cvcam_in_glcam . ob_in_cvcam, ambient light [1,1,1] only (colour = base
# colour = Kd x texture | vertex colour), texture sampler LINEAR / LINEAR_MIPMAP_LINEAR / REPEAT, image uploaded bottom row first,
# colour + GL_DEPTH_COMPONENT/GL_FLOAT read-back, rows flipped, depth linearised -- on the same real GL.  It pins what the numpy
# restatement had to ASSUME about GL (level-of-detail selection, trilinear weights, REPEAT wrap, fill rule, interpolation); the
# statement of pyrender's own scene set-up stays this repo's reading.
_FRAME_VS = """#version 300 es
precision highp float;
in vec3 a_position; in vec3 a_color; in vec2 a_uv;
uniform mat4 u_pv;
out vec3 v_color; out vec2 v_uv;
void main() { gl_Position = u_pv * vec4(a_position, 1.0); v_color = a_color; v_uv = a_uv; }
"""
_FRAME_FS = """#version 300 es
precision highp float;
in vec3 v_color; in vec2 v_uv;
uniform vec3 u_kd; uniform int u_textured; uniform sampler2D u_tex;
out vec4 color;
void main() {
  vec3 base = u_textured != 0 ? texture(u_tex, v_uv).rgb : v_color;
  color = vec4(clamp(base * u_kd, 0.0, 1.0), 1.0);
}
"""


def render_frame_gl(vertices, colors01, faces, ob2cam, K, W, H, uv=None, texture=None, kd=(1.0, 1.0, 1.0), near=0.1, far=2.0,
                    mip_levels=None):
    """rgb uint8 [H,W,3], depth uint16 [H,W] mm as (pyrender depth * 1000).astype(uint16) (predict.py:211).  mip_levels: explicit list
    of levels to upload (tests: the oracle's own pyramid, so that only GL's SAMPLING rules are compared); None = glGenerateMipmap."""
    gl = GL.get()
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    P = np.zeros((4, 4))
    P[0, 0] = 2.0 * fx / W; P[1, 1] = 2.0 * fy / H
    P[0, 2] = 1.0 - 2.0 * cx / W; P[1, 2] = 2.0 * cy / H - 1.0
    P[2, 2] = (far + near) / (near - far); P[2, 3] = 2 * far * near / (near - far); P[3, 2] = -1.0
    V = np.diag([1.0, -1.0, -1.0, 1.0]).dot(ob2cam)
    PV = (P @ V).astype(np.float32)

    def shader(kind, src):
        sh = gl.glCreateShader(kind)
        p = C.c_char_p(src.encode())
        gl.glShaderSource(sh, 1, C.byref(p), None)
        gl.glCompileShader(sh)
        ok = C.c_int()
        gl.glGetShaderiv(sh, GL_COMPILE_STATUS, C.byref(ok))
        assert ok.value, "shader compile failed"
        return sh
    prog = gl.glCreateProgram()
    gl.glAttachShader(prog, shader(GL_VERTEX_SHADER, _FRAME_VS)); gl.glAttachShader(prog, shader(GL_FRAGMENT_SHADER, _FRAME_FS))
    gl.glLinkProgram(prog)
    gl.glUseProgram(prog)
    n = len(vertices)
    data = np.zeros(n, [("a_position", np.float32, 3), ("a_color", np.float32, 3), ("a_uv", np.float32, 2)])
    data["a_position"] = vertices
    data["a_color"] = colors01 if colors01 is not None else 1.0
    data["a_uv"] = uv if uv is not None else 0.0
    vbo, ibo = _VertexBuffer(data), _IndexBuffer(np.asarray(faces).reshape(-1))
    vao = C.c_uint()
    gl.glGenVertexArrays(1, C.byref(vao)); gl.glBindVertexArray(vao)
    gl.glBindBuffer(GL_ARRAY_BUFFER, vbo.id)
    for name in data.dtype.names:
        loc = gl.glGetAttribLocation(prog, name.encode())
        if loc >= 0:
            sub, off = data.dtype.fields[name][0], data.dtype.fields[name][1]
            gl.glEnableVertexAttribArray(loc)
            gl.glVertexAttribPointer(loc, int(np.prod(sub.shape)), GL_FLOAT, 0, data.dtype.itemsize, C.c_void_p(off))
    gl.check("frame program / buffers")
    gl.glUniformMatrix4fv(gl.glGetUniformLocation(prog, b"u_pv"), 1, 1, np.ascontiguousarray(PV).ctypes.data)   # row-major: transpose
    gl.check("u_pv")
    gl.glUniform3f(gl.glGetUniformLocation(prog, b"u_kd"), float(kd[0]), float(kd[1]), float(kd[2]))
    gl.gles.glUniform1i(gl.glGetUniformLocation(prog, b"u_textured"), 1 if texture is not None else 0)
    if texture is not None:
        tid = C.c_uint()
        gl.glGenTextures(1, C.byref(tid))
        gl.gles.glActiveTexture(GL_TEXTURE0)
        gl.glBindTexture(GL_TEXTURE_2D, tid)
        gl.glPixelStorei(GL_UNPACK_ALIGNMENT, 1)
        levels = mip_levels if mip_levels is not None else [np.asarray(texture, np.uint8)]
        for lv, img in enumerate(levels):
            img = np.ascontiguousarray(np.asarray(img, np.uint8)[::-1])               # GL row 0 = bottom row of the image
            gl.glTexImage2D(GL_TEXTURE_2D, lv, GL_RGB8, img.shape[1], img.shape[0], 0, GL_RGB, GL_UNSIGNED_BYTE, img.ctypes.data)
        if mip_levels is None:
            gl.gles.glGenerateMipmap(GL_TEXTURE_2D)
        gl.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR_MIPMAP_LINEAR)
        gl.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR)
        gl.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_REPEAT)
        gl.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_REPEAT)
        gl.gles.glUniform1i(gl.glGetUniformLocation(prog, b"u_tex"), 0)
    gl.check("frame texture")
    old = DEPTH_FORMAT["bits"]
    DEPTH_FORMAT["bits"] = 32
    try:
        # colour attachment RGBA8 renderbuffer-like texture
        ctex = C.c_uint()
        gl.glGenTextures(1, C.byref(ctex))
        gl.gles.glActiveTexture(GL_TEXTURE0 + 1)
        gl.glBindTexture(GL_TEXTURE_2D, ctex)
        gl.glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA8, W, H, 0, GL_RGBA, GL_UNSIGNED_BYTE, None)
        gl.glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST)
        gl.gles.glActiveTexture(GL_TEXTURE0)
        holder = types.SimpleNamespace(id=ctex)
        fbo = _FrameBuffer(holder, _RenderBuffer((H, W)))
        with fbo:
            gl.glEnable(GL_DEPTH_TEST)
            gl.glClearColor(0.0, 0.0, 0.0, 0.0)
            gl.glClear(GL_COLOR_BUFFER_BIT | GL_DEPTH_BUFFER_BIT)
            gl.glViewport(0, 0, W, H)
            gl.check("frame fbo")
            gl.glBindBuffer(GL_ELEMENT_ARRAY_BUFFER, ibo.id)
            gl.glDrawElements(GL_TRIANGLES, int(ibo.data.size), GL_UNSIGNED_INT, None)
            gl.check("frame draw")
            rgb = np.frombuffer(_glReadPixels(0, 0, W, H, GL_RGB, GL_UNSIGNED_BYTE), np.uint8).reshape(H, W, 3)
            z = _glReadPixels(0, 0, W, H, GL_DEPTH_COMPONENT, GL_FLOAT)
    finally:
        DEPTH_FORMAT["bits"] = old
    rgb = rgb[::-1].copy(); z = z[::-1].copy()                                            # pyrender flips on read-back
    hit = z < 1.0
    with np.errstate(divide="ignore", invalid="ignore"):
        z_ndc = z * np.float32(2.0) - np.float32(1.0)
        depth = (np.float32(2.0 * near * far) / (np.float32(far + near) - z_ndc * np.float32(far - near))).astype(np.float32)
    depth[~hit] = 0
    return rgb, (depth * 1000).astype(np.uint16)


def read_mip_levels(texture):
    """The pyramid glGenerateMipmap builds from `texture` (RGB uint8, top row first), read back level by level."""
    gl = GL.get()
    tid = C.c_uint()
    gl.glGenTextures(1, C.byref(tid))
    gl.glBindTexture(GL_TEXTURE_2D, tid)
    gl.glPixelStorei(GL_UNPACK_ALIGNMENT, 1)
    img = np.ascontiguousarray(np.asarray(texture, np.uint8)[::-1])
    rgba = np.concatenate([img, np.full(img.shape[:2] + (1,), 255, np.uint8)], 2)
    rgba = np.ascontiguousarray(rgba)
    gl.glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA8, img.shape[1], img.shape[0], 0, GL_RGBA, GL_UNSIGNED_BYTE, rgba.ctypes.data)
    gl.gles.glGenerateMipmap(GL_TEXTURE_2D)
    out = []
    h, w = img.shape[:2]
    lv = 0
    fb = C.c_uint()
    gl.glGenFramebuffers(1, C.byref(fb))
    while True:
        gl.glBindFramebuffer(GL_FRAMEBUFFER, fb)
        gl.glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, tid, lv)
        buf = np.zeros((h, w, 4), np.uint8)
        gl.glPixelStorei(GL_PACK_ALIGNMENT, 1)
        gl.glReadPixels(0, 0, w, h, GL_RGBA, GL_UNSIGNED_BYTE, buf.ctypes.data)
        gl.check("read mip level %d" % lv)
        out.append(buf[::-1, :, :3].copy())
        if h == 1 and w == 1:
            break
        h, w, lv = max(h // 2, 1), max(w // 2, 1), lv + 1
    gl.glBindFramebuffer(GL_FRAMEBUFFER, 0)
    return out
