/* RaisrVersion.h -- API level tracked: reference v23.11 (Library/RaisrVersion.h:11-17). */
#ifndef RAISR_VERSION_H
#define RAISR_VERSION_H

#define RAISR_VERSION_MAJOR (23)
#define RAISR_VERSION_MINOR (11)
#define RAISR_BACKEND "hip-gfx950"

/* true when the library's API level is at least major.minor */
#define RAISR_CHECK_VERSION(major, minor) \
    ((RAISR_VERSION_MAJOR > (major)) || (RAISR_VERSION_MAJOR == (major) && RAISR_VERSION_MINOR >= (minor)))

#endif
