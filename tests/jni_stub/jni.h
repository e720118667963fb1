/* tests/jni_stub/jni.h — a stand-in for the JDK's <jni.h>, TEST INFRASTRUCTURE ONLY.
 * This image has no JDK; the stub provides just the names kas_jni.cpp uses so that the shim's C
 * side can be compiled and driven from ctypes (tests/test_jni_shim.py).  A "direct ByteBuffer" is
 * a pointer to {address, capacity}. */
#ifndef KAS_TEST_JNI_STUB_H
#define KAS_TEST_JNI_STUB_H
typedef int jint;
typedef long long jlong;
struct kas_stub_buffer { void* address; jlong capacity; };
typedef kas_stub_buffer* jobject;
typedef void* jclass;
struct JNIEnv_ {
  void* GetDirectBufferAddress(jobject b) { return b ? b->address : nullptr; }
  jlong GetDirectBufferCapacity(jobject b) { return b ? b->capacity : -1; }
};
typedef JNIEnv_ JNIEnv;
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#endif
