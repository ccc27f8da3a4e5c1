// Declarations (no definitions) of the GLFW / GLEW / GL names the reference application's main.cpp uses around
// its GI calls (R/main.cpp:23-149), so that main.cpp can be syntax-checked against the facade header on a box
// without a window system (tests/test_drop_in_compile.py).  Test infrastructure; nothing links against this.
#ifndef VCT_TEST_REF_GL_DECLS_H_
#define VCT_TEST_REF_GL_DECLS_H_
struct GLFWwindow;
struct GLFWmonitor;
typedef void (*GLFWframebuffersizefun)(GLFWwindow*, int, int);
typedef void (*GLFWscrollfun)(GLFWwindow*, double, double);
typedef void (*GLFWcursorposfun)(GLFWwindow*, double, double);
typedef unsigned int GLenum;
typedef unsigned int GLbitfield;
typedef unsigned char GLboolean;
extern GLboolean glewExperimental;
enum { GL_TRUE = 1, GLEW_OK = 0, GL_DEPTH_TEST = 0x0B71, GL_LESS = 0x0201, GL_CULL_FACE = 0x0B44, GL_BACK = 0x0405,
       GL_COLOR_BUFFER_BIT = 0x4000, GL_DEPTH_BUFFER_BIT = 0x100 };
enum { GLFW_SAMPLES = 0x2100D, GLFW_CONTEXT_VERSION_MAJOR = 0x22002, GLFW_CONTEXT_VERSION_MINOR = 0x22003,
       GLFW_KEY_ESCAPE = 256, GLFW_KEY_W = 87, GLFW_KEY_S = 83, GLFW_KEY_A = 65, GLFW_KEY_D = 68, GLFW_PRESS = 1,
       GLFW_CURSOR = 0x33001, GLFW_CURSOR_DISABLED = 0x34003 };
int glfwInit(void);
void glfwTerminate(void);
void glfwWindowHint(int, int);
GLFWwindow* glfwCreateWindow(int, int, const char*, GLFWmonitor*, GLFWwindow*);
void glfwMakeContextCurrent(GLFWwindow*);
GLenum glewInit(void);
GLFWframebuffersizefun glfwSetFramebufferSizeCallback(GLFWwindow*, GLFWframebuffersizefun);
GLFWscrollfun glfwSetScrollCallback(GLFWwindow*, GLFWscrollfun);
GLFWcursorposfun glfwSetCursorPosCallback(GLFWwindow*, GLFWcursorposfun);
double glfwGetTime(void);
int glfwWindowShouldClose(GLFWwindow*);
void glfwSetWindowShouldClose(GLFWwindow*, int);
void glfwSwapBuffers(GLFWwindow*);
void glfwPollEvents(void);
int glfwGetKey(GLFWwindow*, int);
void glfwSetInputMode(GLFWwindow*, int, int);
void glEnable(GLenum);
void glDepthFunc(GLenum);
void glCullFace(GLenum);
void glClear(GLbitfield);
void glViewport(int, int, int, int);
#endif
