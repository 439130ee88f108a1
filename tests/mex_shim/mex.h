/* COMPILE-CHECK SHIM ONLY -- not MATLAB's mex.h.  Declares the handful of MEX C
 * API symbols matlab/admm_mex.cpp uses, with the signatures MathWorks documents,
 * so that `g++ -fsyntax-only` can type-check the gateway in an image that has no
 * MATLAB.  Nothing here is linked, shipped or used at run time. */
#ifndef MEX_SHIM_H
#define MEX_SHIM_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
typedef enum { mxDOUBLE_CLASS = 6, mxINT32_CLASS = 12, mxUINT64_CLASS = 15 } mxClassID;
void mexErrMsgIdAndTxt(const char* id, const char* fmt, ...);
void mexWarnMsgIdAndTxt(const char* id, const char* fmt, ...);
int mexAtExit(void (*fn)(void));
mxArray* mxGetField(const mxArray*, mwSize, const char*);
bool mxIsEmpty(const mxArray*);
bool mxIsDouble(const mxArray*);
bool mxIsComplex(const mxArray*);
bool mxIsSparse(const mxArray*);
bool mxIsChar(const mxArray*);
bool mxIsStruct(const mxArray*);
bool mxIsUint64(const mxArray*);
double* mxGetPr(const mxArray*);
void* mxGetData(const mxArray*);
double mxGetScalar(const mxArray*);
size_t mxGetM(const mxArray*);
size_t mxGetN(const mxArray*);
size_t mxGetNumberOfElements(const mxArray*);
mwSize mxGetNumberOfDimensions(const mxArray*);
const mwSize* mxGetDimensions(const mxArray*);
int mxGetString(const mxArray*, char*, mwSize);
mxArray* mxCreateNumericMatrix(mwSize, mwSize, mxClassID, mxComplexity);
mxArray* mxCreateDoubleMatrix(mwSize, mwSize, mxComplexity);
mxArray* mxCreateDoubleScalar(double);
mxArray* mxCreateStructMatrix(mwSize, mwSize, int, const char**);
void mxSetField(mxArray*, mwSize, const char*, mxArray*);
#ifdef __cplusplus
}
#endif
#endif
