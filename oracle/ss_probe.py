"""Black-box probe of the software GL the renderer goldens come from (TEST INFRASTRUCTURE ONLY).

`Probe(W, H)` draws triangles given directly in CLIP coordinates through a pass-through program into an RGBA32F colour
attachment + a 32-bit float depth attachment on SwiftShader (oracle/swiftshader_gl.py) and returns the raw float buffers in
glReadPixels order (row 0 = GL window bottom row).  oracle/ss_rules.py (the arithmetic model of that implementation's fixed
function: sub-pixel snapping, edge walking, plane equations) is fitted and checked against it (tests/test_ss_rules.py)."""
import ctypes as C

import numpy as np

from . import swiftshader_gl as SG

_VS = """#version 300 es
precision highp float;
in vec4 a_pos; in vec4 a_var;
out vec4 v_var;
void main() { gl_Position = a_pos; v_var = a_var; }
"""
_FS = """#version 300 es
precision highp float;
in vec4 v_var;
out vec4 color;
void main() { color = v_var; }
"""
GL_RGBA32F = 0x8814


class Probe:
    def __init__(self, W, H, color_float=True):
        gl = self.gl = SG.GL.get()
        self.W, self.H = W, H
        self.color_float = color_float

        def shader(kind, src):
            sh = gl.glCreateShader(kind)
            p = C.c_char_p(src.encode())
            gl.glShaderSource(sh, 1, C.byref(p), None)
            gl.glCompileShader(sh)
            ok = C.c_int()
            gl.glGetShaderiv(sh, SG.GL_COMPILE_STATUS, C.byref(ok))
            assert ok.value, "shader compile failed"
            return sh
        self.prog = gl.glCreateProgram()
        gl.glAttachShader(self.prog, shader(SG.GL_VERTEX_SHADER, _VS))
        gl.glAttachShader(self.prog, shader(SG.GL_FRAGMENT_SHADER, _FS))
        gl.glLinkProgram(self.prog)
        ok = C.c_int()
        gl.glGetProgramiv(self.prog, SG.GL_LINK_STATUS, C.byref(ok))
        assert ok.value
        self.ctex = C.c_uint()
        gl.glGenTextures(1, C.byref(self.ctex))
        gl.glBindTexture(SG.GL_TEXTURE_2D, self.ctex)
        if color_float:
            gl.glTexImage2D(SG.GL_TEXTURE_2D, 0, GL_RGBA32F, W, H, 0, SG.GL_RGBA, SG.GL_FLOAT, None)
        else:
            gl.glTexImage2D(SG.GL_TEXTURE_2D, 0, SG.GL_RGBA8, W, H, 0, SG.GL_RGBA, SG.GL_UNSIGNED_BYTE, None)
        gl.glTexParameteri(SG.GL_TEXTURE_2D, SG.GL_TEXTURE_MIN_FILTER, SG.GL_NEAREST)
        gl.glTexParameteri(SG.GL_TEXTURE_2D, SG.GL_TEXTURE_MAG_FILTER, SG.GL_NEAREST)
        self.rb = C.c_uint()
        gl.glGenRenderbuffers(1, C.byref(self.rb))
        gl.glBindRenderbuffer(SG.GL_RENDERBUFFER, self.rb)
        gl.glRenderbufferStorage(SG.GL_RENDERBUFFER, SG.GL_DEPTH_COMPONENT32F, W, H)
        self.fbo = C.c_uint()
        gl.glGenFramebuffers(1, C.byref(self.fbo))
        gl.glBindFramebuffer(SG.GL_FRAMEBUFFER, self.fbo)
        gl.glFramebufferTexture2D(SG.GL_FRAMEBUFFER, SG.GL_COLOR_ATTACHMENT0, SG.GL_TEXTURE_2D, self.ctex, 0)
        gl.glFramebufferRenderbuffer(SG.GL_FRAMEBUFFER, SG.GL_DEPTH_ATTACHMENT, SG.GL_RENDERBUFFER, self.rb)
        st = gl.glCheckFramebufferStatus(SG.GL_FRAMEBUFFER)
        assert st == SG.GL_FRAMEBUFFER_COMPLETE, hex(st)
        gl.glBindFramebuffer(SG.GL_FRAMEBUFFER, 0)
        gl.check("probe setup")

    def draw(self, pos, var=None, faces=None, depth_test=True, clear_color=(0.0, 0.0, 0.0, 0.0)):
        """pos [n,4] clip coordinates, var [n,4] one vec4 varying, faces [m,3] -> (colour [H,W,4], depth [H,W] float32)."""
        gl = self.gl
        pos = np.ascontiguousarray(pos, np.float32)
        n = len(pos)
        var = np.zeros((n, 4), np.float32) if var is None else np.ascontiguousarray(var, np.float32)
        faces = np.arange(n, dtype=np.uint32).reshape(-1, 3) if faces is None else np.ascontiguousarray(faces, np.uint32)
        data = np.zeros(n, [("a_pos", np.float32, 4), ("a_var", np.float32, 4)])
        data["a_pos"], data["a_var"] = pos, var
        gl.glUseProgram(self.prog)
        vbo, ibo = SG._VertexBuffer(data), SG._IndexBuffer(faces.reshape(-1))
        vao = C.c_uint()
        gl.glGenVertexArrays(1, C.byref(vao))
        gl.glBindVertexArray(vao)
        gl.glBindBuffer(SG.GL_ARRAY_BUFFER, vbo.id)
        for name in data.dtype.names:
            loc = gl.glGetAttribLocation(self.prog, name.encode())
            if loc >= 0:
                off = data.dtype.fields[name][1]
                gl.glEnableVertexAttribArray(loc)
                gl.glVertexAttribPointer(loc, 4, SG.GL_FLOAT, 0, data.dtype.itemsize, C.c_void_p(off))
        gl.glBindFramebuffer(SG.GL_FRAMEBUFFER, self.fbo)
        if depth_test:
            gl.glEnable(SG.GL_DEPTH_TEST)
        else:
            gl.glDisable(SG.GL_DEPTH_TEST)
        gl.glDisable(SG.GL_CULL_FACE)
        gl.glClearColor(*[float(c) for c in clear_color])
        gl.glClearDepthf(1.0)
        gl.glClear(SG.GL_COLOR_BUFFER_BIT | SG.GL_DEPTH_BUFFER_BIT)
        gl.glViewport(0, 0, self.W, self.H)
        gl.glBindBuffer(SG.GL_ELEMENT_ARRAY_BUFFER, ibo.id)
        gl.glDrawElements(SG.GL_TRIANGLES, int(faces.size), SG.GL_UNSIGNED_INT, None)
        gl.glFinish()
        gl.glPixelStorei(SG.GL_PACK_ALIGNMENT, 1)
        if self.color_float:
            col = np.zeros((self.H, self.W, 4), np.float32)
            gl.glReadPixels(0, 0, self.W, self.H, SG.GL_RGBA, SG.GL_FLOAT, col.ctypes.data)
        else:
            col = np.zeros((self.H, self.W, 4), np.uint8)
            gl.glReadPixels(0, 0, self.W, self.H, SG.GL_RGBA, SG.GL_UNSIGNED_BYTE, col.ctypes.data)
        z = np.zeros((self.H, self.W), np.float32)
        gl.glReadPixels(0, 0, self.W, self.H, SG.GL_DEPTH_COMPONENT, SG.GL_FLOAT, z.ctypes.data)
        gl.check("probe draw")
        gl.glBindFramebuffer(SG.GL_FRAMEBUFFER, 0)
        ids = (C.c_uint * 2)(vbo.id.value, ibo.id.value)
        gl.glDeleteBuffers(2, ids)
        gl.glDeleteVertexArrays(1, C.byref(vao))
        return col, z
