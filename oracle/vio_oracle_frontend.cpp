// placeholder until the front-end restatement lands (keeps `make port` linking)
#include "vio_oracle.h"
